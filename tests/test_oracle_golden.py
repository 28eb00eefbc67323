"""The CPU restatement (oracle/ufo_oracle.cpp) against the golden vectors generated from the
unmodified reference (tests/golden/make_golden.py) and against the known-answer values of
SURVEY.md 8c.  CPU only."""
import numpy as np
import pytest

import golden_util
from oracle import OracleMap


@pytest.mark.parametrize("name", golden_util.names())
def test_port_matches_golden(name, port_available):
    g = golden_util.Golden(name)
    m = OracleMap(kind="port", **g.params)
    for origin, xyz, rgb, kw in g.scans():
        m.insert(origin, xyz, rgb, **kw)
    g.check(m)


def test_kat_single_ray_values(port_available):
    """keys/codes bit-exact, log-odds of hit-then-miss voxel."""
    m = OracleMap(0.16, kind="port")
    m.insert([0.05, 0.05, 0.05], [[1.0, 0.05, 0.05]], max_range=20)
    codes, depths, occ, _ = m.leaves()
    assert codes.tolist() == [246290604621825, 246290604621832, 246290604621833, 246290604621888, 246290604621889, 246290604621896]
    assert depths.tolist() == [0] * 6
    miss = np.float32(np.log(0.4 / 0.6))
    hit = np.float32(np.log(0.7 / 0.3))
    assert np.all(occ[:5] == miss)
    assert occ[5] == np.float32(hit + miss)
    assert abs(float(occ[5]) - 0.441833) < 1e-6
    assert m.last_hits().tolist() == [246290604621896]
    assert m.last_misses().tolist() == codes.tolist()
    assert m.last_steps() == 6


def test_kat_saturation_order_hits_then_misses(port_available):
    """From scan 8 on the hit voxel sits at clamp_max - |miss| = 3.105566: hits clamp, then misses clamp."""
    m = OracleMap(0.16, kind="port")
    for _ in range(10):
        m.insert([0.05, 0.05, 0.05], [[1.0, 0.05, 0.05]], max_range=20)
    _, _, occ, _ = m.leaves()
    assert abs(float(occ[0]) + 2.000028) < 1e-6
    assert abs(float(occ[-1]) - 3.105566) < 1e-6


def test_kat_out_of_range_point_gives_no_hit(port_available):
    m = OracleMap(0.16, kind="port")
    m.insert([0.05, 0.05, 0.05], [[0.05, 30.0, 0.05]], max_range=20)
    _, _, occ, _ = m.leaves()
    assert len(m.last_hits()) == 0
    assert np.all(occ < 0) and len(occ) > 100


def test_kat_discrete_depth1(port_available):
    m = OracleMap(0.16, kind="port")
    m.insert([0.05, 0.05, 0.05], [[1.0, .05, .05], [1.02, .06, .05], [.05, 30, .05]], max_range=20, depth=1, discrete=True)
    codes, depths, occ, _ = m.leaves()
    assert len(codes) == 72
    assert int((depths == 0).sum()) == 8 and int((depths == 1).sum()) == 64
    assert abs(float(occ.max()) - 0.712143) < 1e-6
    assert 30786325577729 in codes[depths == 1].tolist()
    assert len(m.last_hits()) == 1


def test_kat_color_blend(port_available):
    m = OracleMap(0.08, color=True, kind="port")
    o = [0.05, 0.05, 0.05]
    m.insert(o, [[1.0, .05, .05], [1.0, .05, .05]], rgb=[[200, 100, 50], [10, 10, 10]], max_range=20, discrete=True)
    _, _, occ, rgb = m.leaves()
    assert rgb[occ > 0].tolist() == [[200, 100, 50]]
    m.insert(o, [[1.0, .05, .05]], rgb=[[20, 220, 120]], max_range=20, discrete=True)
    _, _, occ, rgb = m.leaves()
    assert rgb[occ > 0].tolist() == [[137, 174, 94]]


def test_constructor_rejects_bad_depth_levels(port_available):
    with pytest.raises(ValueError):
        OracleMap(0.1, depth_levels=1, kind="port")
    with pytest.raises(ValueError):
        OracleMap(0.1, depth_levels=22, kind="port")


def test_empty_cloud_is_a_noop(port_available):
    m = OracleMap(0.16, kind="port")
    m.insert([0, 0, 0], np.zeros((0, 3)), max_range=20, discrete=True)
    assert len(m.leaves()[0]) == 0
    assert len(m.leaves(True)[0]) == 1  # the root: one unknown leaf (OMB:871)


@pytest.mark.parametrize("name", golden_util.server_loop_names())
def test_port_matches_server_loop_golden(name, port_available):
    """Ingest (rosToUfo + Pose6 transform), insertPointCloudDiscrete, robot clearing and point queries of the
    restatement against a recorded run of the unmodified reference (tests/golden/make_golden.py)."""
    import oracle
    from oracle import OracleMap
    g = golden_util.ServerLoop(name)
    m = OracleMap(kind="port", **g.params)
    color = g.params.get("color", False)
    for st in g.steps():
        xyz, rgb = oracle.ingest(st["data"], st["step"], st["off_xyz"], st["off_rgb"] if color else None, st["q"], st["t"], "port")
        assert xyz.shape[0] == st["n_kept"] and golden_util.digest(xyz, rgb) == st["sha_cloud"], "transformed cloud differs"
        m.insert(st["t"], xyz, rgb if color else None, max_range=st["max_range"], discrete=True)
        assert m.clamping_thres()[0] == st["clear_value"]
        m.setValueVolume(st["clear_min"], st["clear_max"], st["clear_value"], st["clear_depth"])
    g.check_map(m)
    g.check_queries(m.query)


@pytest.mark.parametrize("name", ["c1_full", "c2_full_x3", "c4_8poses_x2", "c5_colour_8cm"])
def test_port_matches_reference_digests_full_size(name, port_available):
    """Full-size BASELINE configurations: the port's dumps have the fingerprints the UNMODIFIED reference's dumps had
    when tests/golden/make_digests.py ran (needs neither /root/reference nor oracle/_ref)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_digests import make_scan
    from oracle import OracleMap
    fx = golden_util.digests()[name]
    m = OracleMap(kind="port", **fx["params"])
    for i, ((gen, gkw, ikw), st) in enumerate(zip(fx["scans"], fx["steps"])):
        origin, xyz, rgb = make_scan(gen, gkw)
        m.insert(origin, xyz, rgb, **ikw)
        assert golden_util.dump_digest(m.leaves(True), m.inner()) == tuple(int(v) for v in st["digest"]), f"{name}: scan {i}"


def test_dump_digest_is_order_independent_and_sensitive():
    rng = np.random.default_rng(0)
    n = 1000
    leaves = (rng.integers(0, 1 << 48, n).astype(np.uint64), rng.integers(0, 5, n).astype(np.uint8), rng.normal(size=n).astype(np.float32),
              rng.integers(0, 256, (n, 3)).astype(np.uint8))
    inner = (leaves[0][:100], leaves[1][:100], leaves[2][:100], rng.integers(0, 4, 100).astype(np.uint8), leaves[3][:100])
    d0 = golden_util.dump_digest(leaves, inner)
    perm = rng.permutation(n)
    assert golden_util.dump_digest(tuple(a[perm] for a in leaves), inner) == d0
    l2 = tuple(a.copy() for a in leaves)
    l2[2][17] = np.nextafter(l2[2][17], np.float32(9))
    assert golden_util.dump_digest(l2, inner) != d0
    i2 = tuple(a.copy() for a in inner)
    i2[3][5] ^= 1
    assert golden_util.dump_digest(leaves, i2)[3:] != d0[3:] and golden_util.dump_digest(leaves, i2)[:3] == d0[:3]
