"""Host time inside the insert call (pipelined, HBM-resident clouds): is the host or the GPU the limiter?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
clouds = []
for p in range(8):
    origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(p), seed=100 + p)
    clouds.append((origin, torch.from_numpy(xyz).cuda(), xyz.shape[0]))
g = OccupancyMap(0.16)
for kv in sys.argv[1:]:  # e.g. cast_wgs=248 cast_k=24
    k, v = kv.split("=") if "=" in kv else ("cast_wgs", kv)
    g.set_option(k, int(v))
for rep in range(3):
    for i in range(48):
        o, d, n = clouds[i % 8]
        g.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
g.insertPointCloudWait()
d0 = g.debug()
t0 = time.perf_counter()
N = 480
for i in range(N):
    o, d, n = clouds[i % 8]
    g.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
t1 = time.perf_counter()
g.insertPointCloudWait()
t2 = time.perf_counter()
d1 = g.debug()
h = [(d1[52 + k] - d0[52 + k]) / N / 1000.0 for k in range(4)]
print(f"per scan: wall {1e6 * (t2 - t0) / N:.1f} us (call loop {1e6 * (t1 - t0) / N:.1f} us); inside doInsert: scan enqueue {h[0]:.1f}, map enqueue {h[1]:.1f}, join {h[2]:.1f}, total {h[3]:.1f} us")
