"""Developer aid: the per-code change set after scans at insert depth > 0 (alone, and after scans at depth 0), against the checker."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import OracleMap
from ufomap_amd import OccupancyMap, PointCloud, scans
kind = "reference" if os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "libufo_ref.so")) else "port"
import itertools
for asyn, plan in itertools.product((False, True), ([1], [2], [0, 1], [1, 0], [0, 2, 0], [1, 1], [0, 0, 1, 1])):
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind=kind)
    g.enableChangeDetection(True); o.enableChangeDetection(True)
    res = []
    for i, d in enumerate(plan):
        origin = tuple(np.array(scans.lidar_pose(1)) + [0.05 * i, 0, 0])
        _, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=origin, seed=60 + i)
        g.insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0, d, False, 0, asyn)
        o.insert(origin, xyz, max_range=10.0, discrete=True, depth=d)
        if not asyn or i + 1 == len(plan):
            g.insertPointCloudWait()
        else:
            continue
        a, b = g.changes(), o.changes()
        same_map = all(np.array_equal(x, y) for x, y in zip(g.leaves(True), o.leaves(True)))
        same = len(a[0]) == len(b[0]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        res.append(f"d{d}:{'=' if same else f'DIFF({len(a[0])} vs {len(b[0])})'}{'' if same_map else ' MAP!'}")
        g.resetChangeDetection(); o.resetChangeDetection()
    print(plan, "async" if asyn else "sync", " ".join(res), flush=True)
