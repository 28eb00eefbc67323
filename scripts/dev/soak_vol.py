"""Soak of the volume path's asynchronous form (round 5): a map fed with async=true calls back to back against a map fed synchronously,
frames that differ from call to call (sensor moves, boxes change, the table grows in the middle of walks, plain and colour maps,
pre-growth on and off): the digests must agree at every checkpoint.   python scripts/soak_vol.py [scans]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, OccupancyMapColor, scans
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for color in (False, True):
    for pregrow in (1, 0):
        cls = OccupancyMapColor if color else OccupancyMap
        a, s = cls(resolution=0.004), cls(resolution=0.004)
        a.set_option("vol_pregrow", pregrow); s.set_option("vol_pregrow", pregrow)
        s.set_option("vol_async", 0)
        rng = np.random.default_rng(5 + pregrow + 2 * color)
        keep = []
        bad = 0
        for i in range(n_scans):
            origin = np.array(scans.LIDAR_ORIGIN, dtype=np.float64) + rng.uniform(-0.15, 0.15, 3) * (1.0 if i % 7 else 4.0)
            o, xyz, rgb = scans.rgbd(origin=tuple(origin), seed=100 + i, colored=color, width=320, height=240)
            d = torch.from_numpy(xyz).cuda(); dc = torch.from_numpy(rgb).cuda() if color else None
            keep += [d, dc]
            for m, asy in ((a, True), (s, False)):
                m.insert_device(o, d.data_ptr(), dc.data_ptr() if color else None, xyz.shape[0], 4.0, 0, True, False, 0, asy)
            if i % 10 == 9 or i == n_scans - 1:
                a.insertPointCloudWait()
                same = a.digest() == s.digest()
                bad += 0 if same else 1
                print(f"color={color} pregrow={pregrow} scan {i}: digests {'agree' if same else 'DIFFER'}; volume scans {a.debug()[50]}, growths in walks {a.debug()[49]}", flush=True)
                keep = keep[-4:]
        assert bad == 0
        del a, s
print("soak ok")
