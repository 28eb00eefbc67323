import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMapColor, scans
lo, lx, lc = scans.lidar64(colored=True)
d = torch.from_numpy(lx).cuda(); drgb = torch.from_numpy(lc).cuda()
for wgs in (256, 512, 1024, 2048):
    m = OccupancyMapColor(0.08)
    m.set_option("cast_wgs", wgs)
    for _ in range(3):
        m.insert_device(lo, d.data_ptr(), drgb.data_ptr(), lx.shape[0], 20.0, 0, True)
    m.reset_kernel_times(); m.set_profiling(True)
    for _ in range(5):
        m.insert_device(lo, d.data_ptr(), drgb.data_ptr(), lx.shape[0], 20.0, 0, True)
    m.set_profiling(False)
    kt = m.kernel_times()
    print(wgs, "k_cast_global us:", round(kt["k_cast_global"]["total_ms"] / 5 * 1e3, 1), "fallback passes:", m.debug()[40], "box", [(m.debug()[41] >> 40), (m.debug()[41] >> 20) & 0xfffff, m.debug()[41] & 0xfffff], "ps,pe", m.debug()[42] >> 32, m.debug()[42] & 0xffffffff, "mine", m.debug()[43])
