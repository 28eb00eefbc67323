"""Round 6: one rank's batch steps (ufomap_map_insert_batch over a communicator of ONE rank, real librccl) beside the pipelined single-GPU
path: ms per step, how long the calls take on the host.  python scripts/dev/dev_batch.py [steps]
Under `rocprofv3 --kernel-trace` + scripts/dev/trace_seq.py the kernels of a few steps per hardware queue."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WORLD = int(os.environ.get("WORLD", "1"))  # > 1: one GPU plays rank 0 of WORLD (the all-gather stand-in of scripts/bench_batch_model.py)
if WORLD > 1:
    import subprocess
    shim_src, shim = os.path.join(ROOT, "tests", "cpp", "rccl_shim.cpp"), os.path.join(ROOT, "tests", "cpp", "librccl_shim.so")
    if not os.path.exists(shim) or os.path.getmtime(shim) < os.path.getmtime(shim_src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", shim_src, "-o", shim, "-lrt"], check=True)
    os.environ["UFOMAP_RCCL_LIB"] = os.path.join(ROOT, "tests", "cpp", "librccl_shim.so")
    os.environ["UFOMAP_SHIM_REPLICATE"] = "1"
import numpy as np, torch
from ufomap_amd import OccupancyMap, Comm, scans
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
n = clouds[0][1].shape[0]
d = [torch.from_numpy(c[1]).cuda() for c in clouds]
for mode in (("batch",) if WORLD > 1 else ("single", "batch")):
    m = OccupancyMap(0.16)
    comm = Comm(Comm.unique_id(), WORLD, 0, 0) if mode == "batch" else None
    if comm:
        m.set_option("async_apply", 1)
    for o in sys.argv[2:]:
        k, v = o.split("=")
        m.set_option(k, int(v))
    def step(i):
        p = i % 8
        if comm:
            m.insert_batch(comm, clouds[p][0], d[p].data_ptr(), n, 20.0, 0, True)
        else:
            m.insert_device(clouds[p][0], d[p].data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
    for rep in range(3):
        m.insertPointCloudWait(); m.clear()
        for i in range(16):
            step(i)
        m.insertPointCloudWait(); torch.cuda.synchronize()
        d0 = m.debug(); t0 = time.perf_counter()
        for i in range(16, 16 + K):
            step(i)
        t1 = time.perf_counter()
        m.insertPointCloudWait(); dt = time.perf_counter() - t0
    d1 = m.debug()
    hn = [(d1[52 + k] - d0[52 + k]) / K * 1e-3 for k in range(4)]
    print(f"{mode:7s}: {dt / K * 1e6:7.1f} us per step; calls return after {(t1 - t0) / K * 1e6:6.1f} us; host inside the call {hn[3]:.1f} us (scan half {hn[0]:.1f}, "
          f"{'step incl. scan half' if comm else 'slot'} {hn[1]:.1f}, joins/waits {hn[2]:.1f})" + (f"; counters {comm.counters()}" if comm else ""), flush=True)
    if comm:
        m.set_option("async_apply", 0)
        comm.close()
