#!/bin/bash
# SQ counters for the ray kernel (own pass, --kernel-trace only)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_dda"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-kernels 0"
timeout 300 rocprofv3 -f csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -d "$OUT/a" -o a -- $CMD > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 -f csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE --kernel-trace -d "$OUT/b" -o b -- $CMD > "$OUT/b.log" 2>&1
cd "$ROOT"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_dda/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "k_dda" in r["Kernel_Name"] or "k_walk" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, t) in sorted(acc.items()):
        print(k, "per launch %.0f" % (t / n))
PY
