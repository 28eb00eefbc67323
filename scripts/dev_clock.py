import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
origin, xyz, _ = scans.lidar64()
d = torch.from_numpy(xyz).cuda()
g = OccupancyMap(0.16)
for i in range(9):
    g.insert_device(origin, d.data_ptr(), None, xyz.shape[0], 20.0, 0, discrete=True, async_=False)
    dbg = g.debug()
    st = dbg[10:20]
    print("k_ftail stamps (us since start):", [round((x - st[0]) / 100.0, 2) for x in st])
