import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import OracleMap
from ufomap_amd import OccupancyMap, PointCloud, scans
g, o = OccupancyMap(0.16), OracleMap(0.16, kind="port")
mode = sys.argv[1] if len(sys.argv) > 1 else "dev"
keep = []
for i in range(8):
    origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=tuple(np.array(scans.lidar_pose(1)) + [0.03 * i, 0, 0]), seed=5 + i)
    try:
        if mode == "dev":
            d = torch.from_numpy(xyz).cuda(); keep.append(d)
            g.insert_device(origin, d.data_ptr(), None, len(xyz), 12.0, 0, discrete=True, async_=True)
        elif mode == "sync":
            g.insertPointCloudDiscrete(origin, PointCloud(xyz), 12.0, 0, False, 0, False)
        elif mode == "general":
            g.set_option("fast", 0)
            g.insertPointCloudDiscrete(origin, PointCloud(xyz), 12.0, 0, False, 0, False)
        o.insert(origin, xyz, max_range=12.0, discrete=True)
        g.insertPointCloudWait()
    except Exception as e:
        print(mode, "scan", i, "ERROR", repr(e)); break
    same = all(np.array_equal(a, b) for a, b in zip(g.leaves(True), o.leaves(True)))
    print(mode, "scan", i, "equal" if same else "DIFFERENT", flush=True)
    if not same: break
