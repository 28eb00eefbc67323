"""k_fcast's phases by its own clock stamps (workgroup 0, thread 0; 100 MHz): python scripts/dev/dev_fcast.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d = [torch.from_numpy(c[1]).cuda() for c in clouds]
n = clouds[0][1].shape[0]
m = OccupancyMap(0.16)
m.set_option("ctl_dbg", 1)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    m.set_option(k, int(v))
print("options", sys.argv[1:])
for i in range(24):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True)
names = ["start->lds zeroed", "head loop .. cuts", "walk", "slab store", "tail"]
idx = [30, 31, 35, 36, 37, 38]
if not any(a.startswith("cast_fused=") and a != "cast_fused=2" for a in sys.argv[1:]):
    names = ["start->lds zeroed", "survivors listed", "set-up .. cuts", "walk", "slab store", "tail"]
    idx = [30, 31, 32, 35, 36, 37, 38]
acc = np.zeros(len(names))
for i in range(24, 40):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True)
    g = m.debug()
    st = [g[k] for k in idx]
    acc += np.diff(np.array(st, dtype=np.float64)) * 0.01
print("k_fcast WG 0 phases, us (mean of 16 sync scans):", {k: round(v / 16, 2) for k, v in zip(names, acc)}, "total", round(acc.sum() / 16, 2), "counts", m.last_counts())
m.set_profiling(True); m.reset_kernel_times()
for i in range(40, 56):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True)
kt = m.kernel_times()
print({k: round(v["total_ms"] / max(1, v["launches"]) * 1e3, 1) for k, v in kt.items() if v["launches"]})

# pipelined: the headline loop
m.set_profiling(False)
import time
for rep in range(3):
    m.insertPointCloudWait(); m.clear()
    for i in range(16):
        m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True, False, 0, True)
    m.insertPointCloudWait(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16, 416):
        m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True, False, 0, True)
    m.insertPointCloudWait(); dt = time.perf_counter() - t0
print("pipelined us/scan", round(dt / 400 * 1e6, 2), "Grays/s", round(n / (dt / 400) / 1e9, 3))
g = m.debug()
st = np.array([g[k] for k in idx], dtype=np.float64)
print("slowest lane of workgroup 0, clocks: list read + ray end load + raySetup, reservation + cuts:", [int(g[40 + z]) for z in range(2)], "walk: slowest wave's clocks, longest segment, segments:", int(g[42]), int(g[43]), int(g[39]))
print("k_fcast WG 0 phases of the LAST pipelined scan, us:", {k: round(v, 2) for k, v in zip(names, np.diff(st) * 0.01)}, "total", round((st[-1] - st[0]) * 0.01, 2))
