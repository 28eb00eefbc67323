"""Developer aid (GPU box): the server's per-message sequence (bench.py: server_loop) call by call -- host time of every call
and the kernels behind it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401
from ufomap_amd import OccupancyMap, scans  # noqa: E402

clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
n_pts = clouds[0][1].shape[0]
ident = np.array([1.0, 0.0, 0.0, 0.0])
rec = []
for origin, xyz, _ in clouds:
    b = np.zeros((n_pts, 4), np.float32)
    b[:, :3] = (xyz - np.asarray(origin)[None, :]).astype(np.float32)
    rec.append(np.ascontiguousarray(b).view(np.uint8).reshape(-1))
robot = np.array([0.5, 0.5, 0.75])
m = OccupancyMap(0.16)
for o in sys.argv[1:]:
    k, v = o.split("=")
    m.set_option(k, int(v))
m.enableMinMaxChangeDetection(True)
T = {k: [] for k in ("insert", "setValueVolume", "minmax", "write_ex", "total")}


def step(i, rec_times=True):
    p = i % 8
    o = np.asarray(clouds[p][0])
    t0 = time.perf_counter()
    m.insertPointCloud2(o, ident, rec[p], 16, (0, 4, 8), None, 20.0, 0, True, False, 0, True)
    t1 = time.perf_counter()
    m.setValueVolume(o - robot, o + robot, m.getClampingThresMin(), 0)
    t2 = time.perf_counter()
    mn, mx = m.minmax_change()
    m.resetMinMaxChangeDetection()
    t3 = time.perf_counter()
    data, _ = m.write_ex(aabb=(mn, mx), compress=False, min_depth=0, header=False)
    t4 = time.perf_counter()
    if rec_times:
        for k, a, b in (("insert", t0, t1), ("setValueVolume", t1, t2), ("minmax", t2, t3), ("write_ex", t3, t4), ("total", t0, t4)):
            T[k].append((b - a) * 1e6)
    return len(data)


for i in range(24):
    step(i, False)
for i in range(24, 24 + 80):
    nb = step(i)
print("bytes per publish", nb, " host us per call (median):", {k: round(float(np.median(v)), 1) for k, v in T.items()})
m.reset_kernel_times()
m.set_profiling(True)
for i in range(8):
    step(i, False)
m.set_profiling(False)
kt = m.kernel_times()
for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]):
    if v["launches"]:
        print(f"     {k:24s} {v['total_ms'] / 8 * 1e3:8.1f} us/step  ({v['launches'] / 8:.1f} launches)")
