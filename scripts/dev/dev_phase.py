import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import golden_util
from oracle import OracleMap, available
from ufomap_amd import OccupancyMap, PointCloud, scans
kind = "reference" if available("reference") else "port"
def run(opts, limit=7, sync_only=False, every=False):
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind=kind)
    for k, v in opts.items():
        g.set_option(k, v)
    if limit:
        g.set_option("phase_limit", limit)
    bad = []
    for i in range(14):
        p = [0, 0, 1, 1, 0, 2, 2, 2, 0, 1, 0, 0, 2, 1][i]
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose(p), seed=100 + p)
        discrete = bool(i % 3)
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), 12.0, 0, False, 0, False if sync_only else bool(i % 2))
        o.insert(origin, xyz, max_range=12.0, discrete=discrete)
        if not (every or i % 4 == 3):
            continue
        g.insertPointCloudWait()
        ok = g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())
        d = g.debug()
        bad.append((i, ok, d[51], d[61], g.stats()["bytes"] >> 20))
    print(opts, "limit", limit, "sync" if sync_only else "", [(b[0], b[1]) for b in bad if not b[1]][:3], "resets/fast/MB", bad[-1][2:], flush=True)
run({}, 7)
run({}, 0)
run({}, 7, every=True)
run({"batch_max": 1}, 7)
run({"fast": 0}, 7)
run({"fast": 0, "spec": 0}, 7)
run({"vol": 0}, 7)
run({"lazy_done": 0}, 7)
run({"gates": 0}, 7)
run({"early_map": 0}, 7)
