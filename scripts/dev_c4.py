import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import OracleMap
from ufomap_amd import OccupancyMap, PointCloud, scans
g = OccupancyMap(0.16); o = OracleMap(0.16, kind="port")
for s in range(8):
    origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s)
    g.insertPointCloudDiscrete(origin, PointCloud(xyz), 20.0)
    o.insert(origin, xyz, max_range=20.0, discrete=True)
    gl, ol = g.leaves(True), o.leaves(True)
    gk = {(int(c), int(d)): float(v) for c, d, v in zip(*gl[:3])}
    ok = {(int(c), int(d)): float(v) for c, d, v in zip(*ol[:3])}
    og = sorted(k for k in gk if k not in ok); oo = sorted(k for k in ok if k not in gk)
    print(f"scan {s}: gpu {len(gk)} oracle {len(ok)} only_gpu {len(og)} only_oracle {len(oo)}")
    if og or oo:
        for k in og[:6]: print("  only gpu", k, gk[k])
        for k in oo[:10]: print("  only oracle", k, ok[k])
        # context: for the first gpu-only leaf (a collapsed node), print the oracle's children
        if og:
            c, d = og[0]
            kids = [((c << 3) | i, d - 1) for i in range(8)]
            print("  oracle children of", og[0], [ok.get(k) for k in kids])
            par = (c >> 3, d + 1)
            sib = [((par[0] << 3) | i, d) for i in range(8)]
            print("  siblings in gpu", [gk.get(k) for k in sib])
            print("  siblings in oracle", [ok.get(k) for k in sib])
        break
