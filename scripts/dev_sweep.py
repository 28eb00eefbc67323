"""Dev sweep (one gpurun call): merged vs two-pass map update, ray-walk launch shapes. Prints JSON lines."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufomap_amd import OccupancyMap, scans

origin, xyz, _ = scans.lidar64()
n = xyz.shape[0]
d = torch.from_numpy(xyz).cuda()

def run(opts, steps=300, prof=False, async_=True):
    m = OccupancyMap(0.16)
    for k, v in opts.items():
        m.set_option(k, v)
    for _ in range(30):
        m.insert_device(origin, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
    m.insertPointCloudWait()
    if prof:
        m.reset_kernel_times(); m.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.insert_device(origin, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=async_)
    m.insertPointCloudWait(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    out = {"opts": opts, "ms_per_step": round(dt, 4), "async": async_}
    if prof:
        m.set_profiling(False)
        kt = m.kernel_times()
        out["us"] = {k: round(v["total_ms"] / steps * 1e3, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]) if v["launches"]}
    print(json.dumps(out), flush=True)

run({"cast": 0}, steps=100, prof=True)
run({}, steps=100, prof=True)
run({}, steps=300)
run({}, steps=300, async_=False)
