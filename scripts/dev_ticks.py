"""Dev: in-kernel phase clocks of k_cast (workgroup 0), 100 MHz wall clock."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans, capi
origin, xyz, _ = scans.lidar64()
d = torch.from_numpy(xyz).cuda()
for opts in ({}, {"dda_block": 256}):
    m = OccupancyMap(0.16)
    for k, v in opts.items():
        m.set_option(k, v)
    for _ in range(5):
        m.insert_device(origin, d.data_ptr(), None, xyz.shape[0], 20.0, 0, discrete=True)
    out = (C.c_uint64 * 64)()
    capi.load().ufomap_map_debug(m._h, out, 64)
    t = {i: int(out[40 + i]) for i in range(13)}
    order = [0, 1, 2, 8, 11, 12, 3, 4, 5, 6, 7]
    print(opts, "ticks(10ns):", [(order[i + 1], t[order[i + 1]] - t[order[i]]) for i in range(len(order) - 1)], "total", t[7] - t[0])
