"""Digest fixtures (tests/golden/digests.json) from the UNMODIFIED reference, for configurations whose dumps are
too large to commit or whose CPU run is too long for a test -- above all BASELINE config C3 at insert depth 0
(2 mm, 307 200 points: 85-170 s and ~20 GB on the CPU, 3.4e8 leaves).

Run in the build container (where /root/reference exists):  python tests/golden/make_digests.py [name ...]
Inputs come from the deterministic generators of ufomap_amd/scans.py, so a fixture is (generator arguments,
map parameters, insert arguments) + the order-independent fingerprint of the reference's canonical dumps
(tests/golden_util.dump_digest == ufomap_map_digest of include/ufomap_hip.h) after every listed scan.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from golden_util import dump_digest  # noqa: E402
from oracle import OracleMap, build  # noqa: E402
from ufomap_amd import scans  # noqa: E402

CASES = {
    # name: (map params, [(generator, generator kwargs, insert kwargs), ...])
    "c3_depth0_160x120": (dict(resolution=0.002), [("rgbd", dict(width=160, height=120), dict(max_range=5.0, discrete=True))] * 2),
    # (seven scans of the same frame: the bench times the first into a fresh map, five warm repetitions and an instrumented one,
    # and compares the map after EVERY one of them -- round 4 timed warm scans whose results nobody had checked at full size)
    "c3_depth0_full": (dict(resolution=0.002), [("rgbd", dict(), dict(max_range=5.0, discrete=True))] * 7),
    # coloured 2 mm frames (the reference's published figure is a coloured map at 2 mm): reduced, and at full size (two scans)
    "c3_colour_160x120": (dict(resolution=0.002, color=True), [("rgbd", dict(width=160, height=120, colored=True), dict(max_range=5.0, discrete=True))] * 2),
    "c3_colour_full": (dict(resolution=0.002, color=True), [("rgbd", dict(colored=True), dict(max_range=5.0, discrete=True))] * 3),
    "c1_full": (dict(resolution=0.16), [("lidar64", dict(), dict(max_range=20.0))] * 2),
    "c2_full_x3": (dict(resolution=0.16), [("lidar64", dict(), dict(max_range=20.0, discrete=True))] * 3),
    "c4_8poses_x2": (dict(resolution=0.16), [("lidar64", dict(pose=s % 8, seed=100 + s % 8), dict(max_range=20.0, discrete=True)) for s in range(16)]),
    "c5_colour_8cm": (dict(resolution=0.08, color=True), [("lidar64", dict(pose=3 + s, seed=100 + s, colored=True), dict(max_range=20.0, discrete=True)) for s in range(2)]),
}


def make_scan(gen, kw):
    kw = dict(kw)
    if "pose" in kw:
        kw["origin"] = scans.lidar_pose(kw.pop("pose"))
    return getattr(scans, gen)(**kw)


def main():
    assert build("reference"), "the reference oracle is needed (run where /root/reference exists)"
    path = os.environ.get("UFO_DIGESTS_OUT", os.path.join(HERE, "digests.json"))  # (another file: two generators at a time)
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name in (sys.argv[1:] or list(CASES)):
        params, seq = CASES[name]
        m = OracleMap(kind="reference", **params)
        steps = []
        for gen, gkw, ikw in seq:
            origin, xyz, rgb = make_scan(gen, gkw)
            t0 = time.time()
            m.insert(origin, xyz, rgb, **ikw)
            dt = time.time() - t0
            d = dump_digest(m.leaves(True), m.inner())
            steps.append(dict(digest=[str(v) for v in d], cpu_seconds=round(dt, 2)))
            print(name, len(steps), d[0], d[3], f"{dt:.1f} s", flush=True)
        out[name] = dict(params=params, scans=[[g, k, i] for g, k, i in seq], steps=steps)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
