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
    print("mark A", (dbg[21]-dbg[11])/100.0, "mark B", (dbg[22]-dbg[21])/100.0, "prefix", (dbg[12]-dbg[22])/100.0); st = dbg[10:20]; print("U", dbg[20] & 0xffffffff, "l_end", dbg[20] >> 32)
    print("k_ftail stamps (us since start):", [round((x - st[0]) / 100.0, 2) for x in st])
    print("cut 3a", (dbg[44]-dbg[34])/100.0, "3b+ends", (dbg[35]-dbg[44])/100.0); st = dbg[30:39]
    print("k_fcast wg0 stamps (us):", [round((x - st[0]) / 100.0, 2) for x in st], "mine", dbg[21] if len(dbg) > 21 else None)
