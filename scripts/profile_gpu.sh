#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ (run on the GPU box through gpurun):
#   1. --kernel-trace --stats   per-kernel time of the bench command
#   2. --pmc FETCH_SIZE         and  3. --pmc WRITE_SIZE  in their own runs (TCC slots: 3 + 2 > 4)
# PMC runs use --kernel-trace only (never sys/hip/hsa tracing together with counters).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r01}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-self-check --only-headline --profile-kernels 0 --min-timed-s 0.05 --no-live-traffic"
timeout 600 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 -f csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 -f csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" | head -20
python scripts/summarize_prof.py "$OUT" "$TAG"
# the bandwidth-bound configuration (C3 at insert depth 0, the volume path): per-kernel time and HBM bytes of its kernels
cd /tmp
CMD3="python $ROOT/scripts/dev/dev_c3d0.py"
timeout 600 rocprofv3 -f csv --kernel-trace --stats -d "$OUT/c3_stats" -o stats -- $CMD3 > "$OUT/c3_stats.log" 2>&1
timeout 600 rocprofv3 -f csv --pmc FETCH_SIZE --kernel-trace -d "$OUT/c3_pmc_fetch" -o fetch -- $CMD3 > "$OUT/c3_pmc_fetch.log" 2>&1
timeout 600 rocprofv3 -f csv --pmc WRITE_SIZE --kernel-trace -d "$OUT/c3_pmc_write" -o write -- $CMD3 > "$OUT/c3_pmc_write.log" 2>&1
cd "$ROOT"
python scripts/summarize_prof.py "$OUT" "$TAG" c3
