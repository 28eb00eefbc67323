"""Developer aid (GPU box): BASELINE configs[2] (2 mm RGB-D frame) at insert depth 6 and 3 -- wall time per warm scan and every
kernel behind it (time per scan, launches per scan)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, scans  # noqa: E402

go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
for depth in (6, 3):
    m = OccupancyMap(0.002)
    for o in sys.argv[1:]:
        k, v = o.split("=")
        m.set_option(k, int(v))
    for _ in range(2):
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, depth, True)
    wall = []
    for _ in range(7):
        t0 = time.perf_counter()
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, depth, True)
        wall.append((time.perf_counter() - t0) * 1e3)
    m.set_profiling(True)
    m.reset_kernel_times()
    R = 3
    for _ in range(R):
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, depth, True)
    kt = m.kernel_times()
    m.set_profiling(False)
    rows = sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])
    print("depth", depth, "wall ms median", round(float(np.median(wall)), 3), "sum of kernels", round(sum(v["total_ms"] for _, v in rows) / R, 3), m.last_counts())
    print("   ", {k: (round(v["total_ms"] / R, 3), round(v["launches"] / R, 1)) for k, v in rows})
