"""Developer diagnostic (GPU box): run a few configurations through HIP and the oracle port and
print where they differ instead of asserting. Not part of the test suite."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import OracleMap
from ufomap_amd import OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor, scans


def diff(g, o, tag):
    gl, ol = g.leaves(True), o.leaves(True)
    gk = {(int(c), int(d)): float(v) for c, d, v in zip(*gl[:3])}
    ok = {(int(c), int(d)): float(v) for c, d, v in zip(*ol[:3])}
    only_g = [k for k in gk if k not in ok]
    only_o = [k for k in ok if k not in gk]
    both = [k for k in gk if k in ok]
    bad = [k for k in both if gk[k] != ok[k]]
    gi, oi = g.inner(), o.inner()
    inner_same = all(np.array_equal(a, b) for a, b in zip(gi, oi))
    mm = all(np.array_equal(a, b) for a, b in zip(g.minmax_change(), o.minmax_change()))
    print(f"[{tag}] leaves gpu={len(gk)} oracle={len(ok)} only_gpu={len(only_g)} only_oracle={len(only_o)} value_mismatch={len(bad)} inner_same={inner_same} ({len(gi[0])} vs {len(oi[0])}) aabb_same={mm}")
    for k in (only_g[:5]):
        print("   only gpu   ", k, gk[k])
    for k in (only_o[:5]):
        print("   only oracle", k, ok[k])
    for k in bad[:5]:
        print("   value      ", k, gk[k], ok[k])
    if not inner_same and len(gi[0]) == len(oi[0]):
        for j, nm in enumerate(["codes", "depths", "occ", "flags", "rgb"]):
            d = np.nonzero(np.any(np.atleast_2d(gi[j] != oi[j]).reshape(len(gi[0]), -1), axis=1))[0]
            if len(d):
                i = d[0]
                print(f"   inner {nm} differs at {len(d)} nodes; first: code={gi[0][i]} depth={gi[1][i]} gpu={gi[j][i]} oracle={oi[j][i]}")
    return not (only_g or only_o or bad) and inner_same and mm


def run(tag, params, scan_list, color=False):
    g = (OccupancyMapColor if color else OccupancyMap)(**params)
    o = OracleMap(kind="port", color=color, **params)
    good = True
    for i, (origin, xyz, rgb, kw) in enumerate(scan_list):
        cloud = PointCloudColor(xyz, rgb) if rgb is not None else PointCloud(xyz)
        kw = dict(kw)
        disc = kw.pop("discrete", False)
        t0 = time.time()
        (g.insertPointCloudDiscrete if disc else g.insertPointCloud)(origin, cloud, kw.get("max_range", -1.0), kw.get("depth", 0), kw.get("simple_ray_casting", False))
        t1 = time.time()
        o.insert(origin, xyz, rgb, discrete=disc, **kw)
        t2 = time.time()
        hs = np.array_equal(g.last_hits(), o.last_hits())
        ms = np.array_equal(g.last_misses(), o.last_misses())
        print(f"[{tag}] scan {i}: gpu {1e3*(t1-t0):.2f} ms oracle {1e3*(t2-t1):.1f} ms hits_same={hs} misses_same={ms} counts={g.last_counts()} oracle_steps={o.last_steps()}")
        good &= diff(g, o, f"{tag}#{i}")
    return good


if __name__ == "__main__":
    ok = True
    o0 = np.array([0.05, 0.05, 0.05])
    ok &= run("kat", dict(resolution=0.16), [(o0, np.array([[1.0, .05, .05]]), None, dict(max_range=20.0))] * 3)
    ok &= run("kat_d1", dict(resolution=0.16), [(o0, np.array([[1.0, .05, .05], [1.02, .06, .05], [.05, 30, .05]]), None, dict(max_range=20.0, depth=1, discrete=True))] * 2)
    lo, lx, lc = scans.lidar64(beams=16, azimuths=256, colored=True)
    ok &= run("small_cont", dict(resolution=0.16), [(lo, lx, None, dict(max_range=20.0))] * 2)
    ok &= run("small_disc", dict(resolution=0.16), [(lo, lx, None, dict(max_range=20.0, discrete=True))] * 7)
    ok &= run("small_d2", dict(resolution=0.16), [(lo, lx, None, dict(max_range=8.0, depth=2, discrete=True))] * 3)
    ok &= run("small_color", dict(resolution=0.08), [(lo, lx, lc, dict(max_range=10.0, discrete=True))] * 2, color=True)
    go, gx, gc = scans.rgbd(width=160, height=120, colored=True)
    ok &= run("rgbd_d4", dict(resolution=0.002), [(go, gx, None, dict(max_range=5.0, depth=4, discrete=True))] * 3)
    ok &= run("rgbd_d0_1cm_color", dict(resolution=0.01), [(go, gx, gc, dict(max_range=5.0, discrete=True))] * 2, color=True)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import golden_util
    for nm in ["clip_small_map_continuous", "clip_small_map_discrete", "clip_origin_outside"]:
        gd = golden_util.Golden(nm)
        ok &= run(nm, gd.params, [(o_, x_, r_, kw_) for o_, x_, r_, kw_ in gd.scans()])
    fo, fx, _ = scans.lidar64()
    ok &= run("C1", dict(resolution=0.16), [(fo, fx, None, dict(max_range=20.0))])
    ok &= run("C2", dict(resolution=0.16), [(fo, fx, None, dict(max_range=20.0, discrete=True))] * 3)
    print("ALL OK" if ok else "MISMATCHES")
