"""C3 at insert depth 0 on the volume path: per-kernel times for the k_vdda modes (python scripts/dev/dev_vol.py [modes...])"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
modes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
m = OccupancyMap(0.002)
m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
dig = m.digest()
for mode in modes:
    m.set_option("vol_mode", mode)
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
        ts.append((time.perf_counter() - t0) * 1e3)
    m.set_profiling(True); m.reset_kernel_times()
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True)
    kt = m.kernel_times(); m.set_profiling(False)
    print("mode", mode, "ms", [round(t, 2) for t in ts], {k: round(v["total_ms"], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:6]}, flush=True)
print("counts", m.last_counts(), "debug", m.debug()[48:51], "stats", m.stats())
