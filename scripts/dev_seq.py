"""The bench's moving-sensor sequence: how many scans take the fast path / are repeated, per phase of a repetition."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
clouds = []
for p in range(8):
    origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(p), seed=100 + p)
    clouds.append((origin, torch.from_numpy(xyz).cuda(), xyz.shape[0]))
g = OccupancyMap(0.16)
for rep in range(2):
    g.insertPointCloudWait(); g.clear()
    prev = g.debug()
    for i in range(48):
        o, d, n = clouds[i % 8]
        g.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
        if i in (7, 15, 23, 47):
            g.insertPointCloudWait()
            d1 = g.debug()
            print(f"rep {rep} scans ..{i}: fast +{d1[61]-prev[61]} spec +{d1[62]-prev[62]} redo +{d1[63]-prev[63]}", flush=True)
            prev = d1
    t0 = time.perf_counter()
    for i in range(48, 96):
        o, d, n = clouds[i % 8]
        g.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True, async_=True)
    g.insertPointCloudWait()
    d1 = g.debug()
    print(f"rep {rep} 48 more scans: {(time.perf_counter()-t0)/48*1e6:.1f} us/scan, fast +{d1[61]-prev[61]} redo +{d1[63]-prev[63]}")
