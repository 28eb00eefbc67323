#!/bin/bash
# the two --pmc passes of scripts/profile_gpu.sh alone (own runs, --kernel-trace only), with short timeouts
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r02}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-self-check --only-headline --profile-kernels 0 --min-timed-s 0.05"
timeout 150 rocprofv3 -f csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 150 rocprofv3 -f csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
cd "$ROOT"
python scripts/summarize_prof.py "$OUT" "$TAG" > /dev/null
