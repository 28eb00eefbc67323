import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufomap_amd import OccupancyMap, scans
origin, xyz, _ = scans.lidar64()
d = torch.from_numpy(xyz).cuda()
m = OccupancyMap(0.16)
for i in range(12):
    m.insert_device(origin, d.data_ptr(), None, xyz.shape[0], 20.0, 0, True)
out = np.zeros(64, np.uint64)
m._lib.ufomap_map_debug(m._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), 64)
for ph in (0, 1):
    st = out[ph*32:ph*32+32].astype(np.int64)
    lv = [(l, st[l]) for l in range(2, 17) if st[l]]
    print("phase", ph, "levels", [l for l,_ in lv], "dt(us) per level", [round((b[1]-a[1])/100.0,2) for a,b in zip(lv, lv[1:]+[(31, st[31])])])
