"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
Each case stores its inputs (params, per-scan origin/xyz/rgb) and the reference's canonical dumps:
the known (non-unknown) leaves in full, SHA-256 digests of the complete leaf dump (incl. unknown
leaves) and of the inner-node dump (value, contains_free/contains_unknown flags, colour), and the
min/max change AABB, and the SHA-256 of the map's byte stream as the reference's own write() produces
it (octree.h:833-868).  The fixtures are self-contained: neither /root/reference nor oracle/_ref is
needed to check against them.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import OracleMap, build  # noqa: E402
from ufomap_amd import scans  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def clean_rays(params, origin, xyz, **kw):
    """Keep the points whose ray stays inside the key range [0, 2^L) after clipping.

    When moveLineInside puts an end point onto (or an ulp over) a face of the map cube, toKey gives
    2^L or -1; the reference then aliases the key to the opposite side of the map (or walks ~2^31
    cells). That behaviour is outside the parity contract, so the clipping fixtures use
    only rays that are clipped but keep all keys in range. The port reports both conditions."""
    from oracle import RunawayRay
    keep = []
    for i in range(len(xyz)):
        m = OracleMap(kind="port", **params)
        try:
            m.insert(origin, xyz[i:i + 1], **kw)
        except RunawayRay:
            continue
        if m.last_oob() == 0:
            keep.append(i)
    return xyz[np.array(keep)]


def cases():
    o = np.array([0.05, 0.05, 0.05])
    kat = np.array([[1.0, 0.05, 0.05]])
    yield "kat_single_ray", dict(resolution=0.16), [dict(origin=o, xyz=kat, max_range=20.0)] * 1
    yield "kat_saturation_x10", dict(resolution=0.16), [dict(origin=o, xyz=kat, max_range=20.0)] * 10
    d1 = np.array([[1.0, .05, .05], [1.02, .06, .05], [.05, 30, .05]])
    yield "kat_discrete_depth1", dict(resolution=0.16), [dict(origin=o, xyz=d1, max_range=20.0, depth=1, discrete=True)]
    yield "kat_color", dict(resolution=0.08, color=True), [
        dict(origin=o, xyz=np.array([[1.0, .05, .05], [1.0, .05, .05]]), rgb=np.array([[200, 100, 50], [10, 10, 10]], np.uint8), max_range=20.0, discrete=True),
        dict(origin=o, xyz=kat, rgb=np.array([[20, 220, 120]], np.uint8), max_range=20.0, discrete=True)]
    lo, lx, lc = scans.lidar64(beams=8, azimuths=128, colored=True)
    yield "lidar_small_continuous", dict(resolution=0.16), [dict(origin=lo, xyz=lx, max_range=20.0)]
    yield "lidar_small_discrete", dict(resolution=0.16), [dict(origin=lo, xyz=lx, max_range=20.0, discrete=True)]
    yield "lidar_small_discrete_x6", dict(resolution=0.16), [dict(origin=lo, xyz=lx, max_range=20.0, discrete=True)] * 6
    yield "lidar_small_range8_depth2", dict(resolution=0.16), [dict(origin=lo, xyz=lx, max_range=8.0, depth=2, discrete=True)] * 2
    yield "lidar_small_simple", dict(resolution=0.16), [dict(origin=lo, xyz=lx, max_range=12.0, simple_ray_casting=True)]
    yield "lidar_small_nopruning", dict(resolution=0.16, automatic_pruning=False), [dict(origin=lo, xyz=lx, max_range=20.0, discrete=True)] * 3
    yield "lidar_small_color_8cm", dict(resolution=0.08, color=True), [dict(origin=lo, xyz=lx, rgb=lc, max_range=10.0, discrete=True)] * 2
    seq = []
    for s in range(4):
        so, sx, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, beams=8, azimuths=128)
        seq.append(dict(origin=so, xyz=sx, max_range=20.0, discrete=True))
    yield "lidar_small_4poses", dict(resolution=0.16), seq
    ro, rx, _ = scans.random_cloud(500, seed=7, extent=30.0)
    small = dict(resolution=0.5, depth_levels=6)
    yield "clip_small_map_continuous", small, [dict(origin=ro, xyz=clean_rays(small, ro, rx, max_range=-1.0), max_range=-1.0)]
    yield "clip_small_map_discrete", small, [dict(origin=ro, xyz=clean_rays(small, ro, rx, max_range=25.0, discrete=True, depth=1), max_range=25.0, discrete=True, depth=1)]
    oo = np.array([40.0, 3.0, -2.0])
    yield "clip_origin_outside", small, [dict(origin=oo, xyz=clean_rays(small, oo, rx, max_range=-1.0), max_range=-1.0)]
    go, gx, gc = scans.rgbd(width=40, height=30, colored=True)
    yield "rgbd_small_2mm_depth4", dict(resolution=0.002), [dict(origin=go, xyz=gx, max_range=5.0, depth=4, discrete=True)]
    yield "rgbd_small_2cm_color", dict(resolution=0.02, color=True), [dict(origin=go, xyz=gx, rgb=gc, max_range=5.0, discrete=True)] * 2
    yield "sensor_model_custom", dict(resolution=0.1, depth_levels=14, occupied_thres=0.6, free_thres=0.35, prob_hit=0.8, prob_miss=0.3,
                                      clamping_thres_min=0.05, clamping_thres_max=0.99), [dict(origin=lo, xyz=lx, max_range=15.0, discrete=True)] * 4


def server_loop_cases():
    """The server's per-message sequence (ufomap_mapping/src/server.cpp:113-168): rosToUfo + transform (raw
    PointCloud2 records + pose), insertPointCloudDiscrete, robot clearing -- then point queries."""
    for name, color, res, clear_depth in (("srv_loop_16cm", False, 0.16, 0), ("srv_loop_color_8cm_clear1", True, 0.08, 1)):
        rng = np.random.default_rng(11)
        steps = []
        for s in range(3):
            _, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=(0.0, 0.0, 0.0), seed=40 + s, colored=True)
            n, step = xyz.shape[0], 32
            buf = rng.integers(0, 256, (n, step), dtype=np.uint8)
            f = xyz.astype(np.float32)
            f[::23, s % 3] = np.nan
            buf[:, 0:12] = f.view(np.uint8).reshape(n, 12)
            buf[:, 16], buf[:, 17], buf[:, 18] = rgb[:, 2], rgb[:, 1], rgb[:, 0]
            q = np.array([np.cos(0.15 * (s + 1)), 0.01 * s, -0.02, np.sin(0.15 * (s + 1))])
            q /= np.linalg.norm(q)
            t = np.array([0.5 * s, -0.25 * s, 0.8])
            steps.append(dict(data=buf, step=step, off_xyz=(0, 4, 8), off_rgb=(18, 17, 16), q=q, t=t, max_range=8.0,
                              clear_min=t - np.array([0.4, 0.4, 0.5]), clear_max=t + np.array([0.4, 0.4, 0.5]), clear_depth=clear_depth))
        qs = np.concatenate([rng.uniform(-9, 9, (3000, 3)), rng.uniform(-3000, 3000, (200, 3))])
        yield name, dict(resolution=res, color=color), steps, qs


def make_server_loops(index):
    import oracle
    for name, params, steps, qs in server_loop_cases():
        m = OracleMap(kind="reference", **params)
        arrays, meta = {"queries": qs}, dict(params=params, steps=[])
        for i, st in enumerate(steps):
            orgb = st["off_rgb"] if params["color"] else None
            xyz, rgb = oracle.ingest(st["data"], st["step"], st["off_xyz"], orgb, st["q"], st["t"], "reference")
            m.insert(st["t"], xyz, rgb if params["color"] else None, max_range=st["max_range"], discrete=True)
            thr = m.clamping_thres()[0]
            m.setValueVolume(st["clear_min"], st["clear_max"], thr, st["clear_depth"])
            arrays[f"d{i}_data"] = st["data"]
            arrays[f"d{i}_q"], arrays[f"d{i}_t"] = st["q"], st["t"]
            arrays[f"d{i}_clear_min"], arrays[f"d{i}_clear_max"] = st["clear_min"], st["clear_max"]
            meta["steps"].append(dict(step=st["step"], off_xyz=st["off_xyz"], off_rgb=st["off_rgb"], max_range=st["max_range"],
                                      clear_depth=st["clear_depth"], clear_value=thr, n_kept=int(xyz.shape[0]),
                                      sha_cloud=digest(xyz, rgb)))
        lc, ld, lv, lrgb = m.leaves(True)
        ic, idp, iv, ifl, irgb = m.inner()
        wb = m.write()
        meta.update(n_leaves_all=int(len(lc)), n_inner=int(len(ic)), sha_leaves_all=digest(lc, ld, lv, lrgb),
                    sha_inner=digest(ic, idp, iv, ifl, irgb), write_size=len(wb), sha_write=hashlib.sha256(wb).hexdigest(),
                    query_depths=[0, 1, 3])
        for d in meta["query_depths"]:
            lo, stt = m.query(qs, d)
            arrays[f"q{d}_logodds"], arrays[f"q{d}_state"] = lo, stt
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
        index[name] = dict(leaves=int(len(lc)), inner=int(len(ic)))
        print(f"{name:32s} leaves={len(lc):7d} inner={len(ic):6d}")


def main():
    assert build("reference"), "needs /root/reference to build oracle/_ref"
    index = {}
    make_server_loops(index)
    for name, params, scan_list in cases():
        m = OracleMap(kind="reference", **params)
        arrays = {}
        meta = dict(params=params, scans=[])
        for i, sc in enumerate(scan_list):
            sc = dict(sc)
            origin, xyz, rgb = sc.pop("origin"), sc.pop("xyz"), sc.pop("rgb", None)
            m.insert(origin, xyz, rgb, **sc)
            key = None
            for j in range(i):  # de-duplicate repeated identical scans
                if scan_list[j] is scan_list[i]:
                    key = meta["scans"][j]["data"]
            if key is None:
                key = f"s{i}"
                arrays[key + "_origin"] = np.asarray(origin, np.float64)
                arrays[key + "_xyz"] = np.asarray(xyz, np.float64)
                if rgb is not None:
                    arrays[key + "_rgb"] = np.asarray(rgb, np.uint8)
            meta["scans"].append(dict(data=key, has_rgb=rgb is not None, kwargs=sc))
        lc, ld, lv, lrgb = m.leaves(True)
        ic, idp, iv, ifl, irgb = m.inner()
        mn, mx = m.minmax_change()
        kc, kd, kv, krgb = m.leaves(False)
        arrays.update(leaf_codes=kc, leaf_depths=kd, leaf_occ=kv, leaf_rgb=krgb, min_change=mn, max_change=mx)
        wb = m.write()
        meta.update(write_size=len(wb), sha_write=hashlib.sha256(wb).hexdigest())
        meta.update(n_leaves_all=int(len(lc)), n_inner=int(len(ic)),
                    sha_leaves_all=digest(lc, ld, lv, lrgb), sha_inner=digest(ic, idp, iv, ifl, irgb))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
        index[name] = dict(leaves=int(len(lc)), known=int(len(kc)), inner=int(len(ic)))
        print(f"{name:32s} leaves={len(lc):7d} known={len(kc):7d} inner={len(ic):6d}")
    with open(os.path.join(HERE, "INDEX.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
