"""Dev: is the pipelined insert host-bound? Per-call host time vs step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
origin, xyz, _ = scans.lidar64()
n = xyz.shape[0]
d = torch.from_numpy(xyz).cuda()
import sys as _s
m = OccupancyMap(0.16)
for kv in _s.argv[1:]:
    k, v = kv.split("=")
    m.set_option(k, int(v))
for _ in range(30):
    m.insert_device(origin, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
m.insertPointCloudWait()
ts = []
t0 = time.perf_counter()
for _ in range(400):
    a = time.perf_counter()
    m.insert_device(origin, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
    ts.append(time.perf_counter() - a)
m.insertPointCloudWait()
tot = (time.perf_counter() - t0) / 400
ts = np.array(ts) * 1e6
print("step us", round(tot * 1e6, 1), "call us: median", round(float(np.median(ts)), 1), "p10", round(float(np.percentile(ts, 10)), 1), "p90", round(float(np.percentile(ts, 90)), 1))
# how long does the GPU need when everything is queued? enqueue-only cost: time N calls with spec but without waiting... (call includes the join)
print("replays / spec used / repeats:", m.debug()[61:64])
