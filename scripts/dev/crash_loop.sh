#!/bin/bash
# Developer aid (GPU box): the driver's bench command N times in a short form, counting runs that do not exit with 0 (round 6: a use-after-free
# behind the batch_step_n1 leg killed one run in eight).  bash scripts/dev/crash_loop.sh [runs=40]
mkdir -p gpurun_out/crash; ulimit -c 0
n=0
for i in $(seq 1 ${1:-40}); do
  python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-self-check --min-timed-s 0.1 > gpurun_out/crash/s.json 2> gpurun_out/crash/s.err; rc=$?
  if [ $rc -ne 0 ]; then n=$((n+1)); echo run $i rc=$rc; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^Extension\|^$" gpurun_out/crash/s.err | head -n 8; fi
done
echo crashes $n of ${1:-40}
