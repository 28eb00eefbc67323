"""C3 at insert depth 0, the FRESH scan: what its time is made of (allocations, memsets, creations): per-kernel times of the first
scan into a new map, of the first scan after clear() (buffers kept), and the host's laps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans, capi
go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
def one(m, tag, prof):
    if prof:
        m.set_profiling(True); m.reset_kernel_times()
    a0 = capi.alloc_counters()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
    ms = (time.perf_counter() - t0) * 1e3
    a1 = capi.alloc_counters()
    kt = m.kernel_times() if prof else {}
    if prof:
        m.set_profiling(False)
    print(tag, "ms", round(ms, 2), "allocs", {k: a1[k] - a0[k] for k in a1}, {k: round(v["total_ms"], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:10]}, flush=True)
m = OccupancyMap(0.002); one(m, "fresh map, no events ", False)
m.clear(); one(m, "after clear, no events", False)
m.clear(); one(m, "after clear, events   ", True)
m2 = OccupancyMap(0.002); one(m2, "fresh map, events     ", True)
one(m2, "warm                  ", True)
