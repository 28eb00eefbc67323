"""Developer aid (GPU box): per-kernel times of scans whose ray grid is beyond LDS (the general path)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, OccupancyMapColor, scans  # noqa: E402

lo, lx, lc = scans.lidar64(colored=True)
d = torch.from_numpy(lx).cuda()
drgb = torch.from_numpy(lc).cuda()
CASES = (("colour 8 cm", OccupancyMapColor, 0.08, True, {}), ("plain 8 cm", OccupancyMap, 0.08, False, {}), ("plain 16 cm", OccupancyMap, 0.16, False, {}),
         ("colour 16 cm", OccupancyMapColor, 0.16, True, {}), ("colour 16 cm, general path", OccupancyMapColor, 0.16, True, {"fast_color": 0}))
for name, cls, res, rgb, opts in CASES:
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    m = cls(res)
    for k, v in opts.items():
        m.set_option(k, v)
    for o in sys.argv[1:]:
        k, v = o.split("=")
        m.set_option(k, int(v))
    ins = lambda: m.insert_device(lo, d.data_ptr(), drgb.data_ptr() if rgb else None, lx.shape[0], 20.0, 0, True)  # noqa: E731
    for _ in range(4):
        ins()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        ins()
        ts.append(time.perf_counter() - t0)
    m.reset_kernel_times()
    m.set_profiling(True)
    for _ in range(5):
        ins()
    m.set_profiling(False)
    kt = m.kernel_times()
    tot = sum(v["total_ms"] for v in kt.values()) / 5 * 1e3
    dbg = m.debug()
    print(f"--- {name}: sync insert median {np.median(ts) * 1e6:.0f} us; kernels {tot:.0f} us over {sum(v['launches'] for v in kt.values()) / 5:.0f} launches; fast-path scans {dbg[61]}")
    for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]):
        if v["launches"]:
            print(f"     {k:24s} {v['total_ms'] / 5 * 1e3:8.1f} us/scan  ({v['launches'] / 5:.0f} launches)")
    del m
