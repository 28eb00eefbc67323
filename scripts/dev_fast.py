"""Bring-up check of the fast path (fast_kernels.h) against the oracle: a few scans, compared after each one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import OracleMap, available
from ufomap_amd import OccupancyMap, PointCloud, scans

kind = "reference" if available("reference") else "port"
small = "--full" not in sys.argv
kw = dict(beams=16, azimuths=512) if small else {}
for discrete in (True, False):
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind="port")
    if "--memset" in sys.argv:
        g.set_option("fast", 2)
    if "--dbg" in sys.argv:
        g.set_option("fast", 3)
    poses = [0, 0, 0, 1, 1, 2, 0, 3, 3, 3, 3, 3, 3, 3, 1, 2] if "--static" not in sys.argv else [0] * 8
    for i, p in enumerate(poses):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(p), seed=100 + p, **kw)
        fn = g.insertPointCloudDiscrete if discrete else g.insertPointCloud
        fn(origin, PointCloud(xyz), 20.0, 0, False, 0, False)
        o.insert(origin, xyz, max_range=20.0, discrete=discrete)
        d = g.debug()
        gl, ol = g.leaves(True), o.leaves(True)
        gi, oi = g.inner(), o.inner()
        okl = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(gl, ol))
        oki = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(gi, oi))
        okh = np.array_equal(g.last_hits(), o.last_hits())
        okm = np.array_equal(g.last_misses(), o.last_misses())
        c = g.last_counts()
        print(f"discrete={discrete} scan {i} pose {p}: fast={d[61]} spec={d[62]} redo={d[63]} leaves {len(gl[0])}/{len(ol[0])} ok={okl} inner {len(gi[0])}/{len(oi[0])} ok={oki} hits={okh} misses={okm} steps={c['steps']}/{o.last_steps()} rays={c['rays']}/{len(o.last_rays())}", flush=True)
        if not (okl and oki):
            if len(gl[0]) == len(ol[0]):
                bad = np.nonzero((gl[0] != ol[0]) | (gl[1] != ol[1]) | (gl[2] != ol[2]))[0]
                print("  leaf diffs:", len(bad), [(int(gl[0][k]), int(gl[1][k]), float(gl[2][k]), int(ol[0][k]), int(ol[1][k]), float(ol[2][k])) for k in bad[:5]])
            else:
                sg = set(zip(gl[1].tolist(), gl[0].tolist())); so = set(zip(ol[1].tolist(), ol[0].tolist()))
                print("  only gpu:", sorted(sg - so)[:6], " only oracle:", sorted(so - sg)[:6])
            if len(gi[0]) == len(oi[0]):
                bad = np.nonzero((gi[0] != oi[0]) | (gi[1] != oi[1]) | (gi[2] != oi[2]) | (gi[3] != oi[3]))[0]
                print("  inner diffs:", len(bad), [(int(gi[0][k]), int(gi[1][k]), float(gi[2][k]), int(gi[3][k]), float(oi[2][k]), int(oi[3][k])) for k in bad[:5]])
            sys.exit(1)
    # pipelined
    g2 = OccupancyMap(0.16)
    for i, p in enumerate(poses):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(p), seed=100 + p, **kw)
        fn = g2.insertPointCloudDiscrete if discrete else g2.insertPointCloud
        fn(origin, PointCloud(xyz), 20.0, 0, False, 0, True)
    g2.insertPointCloudWait()
    d = g2.debug()
    same = all(np.array_equal(a, b) for a, b in zip(g2.leaves(True), g.leaves(True))) and all(np.array_equal(a, b) for a, b in zip(g2.inner(), g.inner()))
    print(f"discrete={discrete} pipelined: fast={d[61]} spec={d[62]} redo={d[63]} same={same}", flush=True)
    if not same:
        sys.exit(2)
print("dev_fast ok")
