import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufomap_amd import OccupancyMap, scans
go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
for depth in (6, 3):
    m = OccupancyMap(0.002)
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, depth, True)
    m.set_profiling(True); m.reset_kernel_times()
    for _ in range(3):
        m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, depth, True)
    kt = m.kernel_times()
    print("depth", depth, {k: round(v["total_ms"]/3, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}, m.last_counts())
