"""k_ftail's phases by its own clock stamps (ufomap_map_debug words 10..22, 100 MHz), sync scans on the steady-state path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d = [torch.from_numpy(c[1]).cuda() for c in clouds]
m = OccupancyMap(0.16)
m.set_option("ctl_dbg", 1)
for i in range(24):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, clouds[0][1].shape[0], 20.0, 0, True)
names = {10: "start", 11: "lds cleared", 21: "tiles marked", 22: "ancestors marked", 12: "prefix", 13: "blocks found/loaded", 14: "inherit", 15: "tile records", 16: "last-update recs",
         17: "wide levels", 18: "narrow levels", 19: "written back", 23: "reported to the host"}
order = [10, 11, 21, 22, 12, 13, 14, 15, 16, 17, 18, 19, 23]
acc = np.zeros(len(order) - 1)
n = 0
for i in range(24, 48):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, clouds[0][1].shape[0], 20.0, 0, True)
    dbg = m.debug()
    st = [dbg[k] for k in order]
    if all(st):
        acc += np.diff(np.array(st, dtype=np.float64)) / 100.0
        n += 1
print("k_ftail phases, us (mean of %d sync scans):" % n)
for k, v in zip(order[1:], acc / max(n, 1)):
    print(f"  -> {names[k]:24s} {v:6.2f}")
dbg = m.debug()
print("  narrow levels by step8 / in runs:", dbg[24] & 0xFFFFFFFF, dbg[24] >> 32, "clocks:", dbg[25], dbg[26])
print("  total", round(float(acc.sum() / max(n, 1)), 2), "nodes U, last level", m.debug()[20] & 0xFFFFFFFF, m.debug()[20] >> 32)
m.set_profiling(True); m.reset_kernel_times()
for i in range(48, 72):
    m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, clouds[0][1].shape[0], 20.0, 0, True)
kt = m.kernel_times()
print({k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in kt.items() if v["launches"]})
