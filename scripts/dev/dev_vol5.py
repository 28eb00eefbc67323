"""C3 at insert depth 0 on the volume path, round 5: the segmented ray walk against the one-lane-per-ray kernel, segment length and
launch shape sweeps; per-kernel times by HIP events.   python scripts/dev/dev_vol5.py [quick]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
m = OccupancyMap(0.002)
torch.cuda.synchronize(); t0 = time.perf_counter()
m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
print("fresh ms", round((time.perf_counter() - t0) * 1e3, 2), flush=True)
dig1 = m.digest()
combos = [dict(vol_mode=16), dict(vol_mode=0, vol_seg=192, vol_walk_blocks=384)]
if "quick" not in sys.argv:
    combos += [dict(vol_seg=96), dict(vol_seg=128), dict(vol_seg=256), dict(vol_seg=384), dict(vol_seg=192, vol_walk_blocks=256), dict(vol_seg=192, vol_walk_blocks=768),
               dict(vol_seg=192, vol_walk_blocks=1536), dict(vol_seg=192, vol_walk_blocks=384, vol_mode=8)]
for c in combos:
    for k, v in c.items():
        m.set_option(k, v)
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
        ts.append((time.perf_counter() - t0) * 1e3)
    m.set_profiling(True); m.reset_kernel_times()
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
    kt = m.kernel_times(); m.set_profiling(False)
    print(c, "ms", [round(t, 2) for t in ts], {k: round(v["total_ms"], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}, "steps", m.last_counts()["steps"], flush=True)
print("counts", m.last_counts(), "debug", m.debug()[48:51], "stats", m.stats())
# pipelined: async calls back to back (the walk of scan i overlaps the ray casting of scan i + 1)
for asy in (0, 1):
    m.set_option("vol_async", asy)
    for _ in range(4):  # (each hand-over set allocates and clears brick grids of its own when it is first used)
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True, False, 0, True)
    m.insertPointCloudWait()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True, False, 0, True)
    m.insertPointCloudWait()
    print("vol_async", asy, "ms per scan over 8 async calls", round((time.perf_counter() - t0) * 1e3 / 8, 3), flush=True)
