"""Developer aid: an asynchronous steady-state scan followed at once by an asynchronous scan with fixed-step casting / early stopping."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import OracleMap
from ufomap_amd import OccupancyMap, PointCloud, scans
for simple in (False, True):
    for es in (0, 1, 2, 3):
        for first_async in (True, False):
            g, o = OccupancyMap(0.16), OracleMap(0.16, kind="port")
            pose = np.array(scans.lidar_pose(1))
            ok = True
            for i in range(6):
                origin = tuple(pose + [0.03 * i, 0, 0])
                _, xyz, _ = scans.lidar64(beams=16, azimuths=128 if i % 2 else 512, origin=origin, seed=40 + i)
                if i in (2, 4):
                    g.insertPointCloudDiscrete(origin, PointCloud(xyz), 12.0, 0, simple, es, True)
                    o.insert(origin, xyz, max_range=12.0, discrete=True, simple_ray_casting=simple, early_stopping=es)
                else:
                    g.insertPointCloudDiscrete(origin, PointCloud(xyz), 12.0, 0, False, 0, first_async)
                    o.insert(origin, xyz, max_range=12.0, discrete=True)
            g.insertPointCloudWait()
            same = all(np.array_equal(a, b) for a, b in zip(g.leaves(True), o.leaves(True)))
            print(f"simple={simple} early_stopping={es} other scans async={first_async}: {'equal' if same else 'DIFFERENT'}", flush=True)
