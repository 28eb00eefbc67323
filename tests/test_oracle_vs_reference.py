"""Pin the CPU restatement against the unmodified reference (oracle/_ref/libufo_ref.so) on inputs
beyond the golden fixtures.  Skipped where neither /root/reference nor a prebuilt _ref exists."""
import numpy as np
import pytest

from conftest import same_dump
from oracle import OracleMap, RunawayRay
from ufomap_amd import scans


def _pair(**params):
    return OracleMap(kind="reference", **params), OracleMap(kind="port", **params)


def _assert_same(a, b):
    assert same_dump(a.leaves(True), b.leaves(True)), "leaf dump differs"
    assert same_dump(a.inner(), b.inner()), "inner dump differs"
    assert same_dump(a.minmax_change(), b.minmax_change()), "change AABB differs"
    assert a.write() == b.write(), "byte stream (Octree::write) differs"


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", ["continuous", "discrete", "discrete_d2", "simple"])
def test_random_clouds_small_map(seed, mode, ref_available, port_available):
    """Small map cube (+-16 m) and +-30 m points: exercises moveLineInside clipping on both ends."""
    if not ref_available:
        pytest.skip("reference build not available")
    a, b = _pair(resolution=0.5, depth_levels=6)
    kw = dict(continuous=dict(), discrete=dict(discrete=True), discrete_d2=dict(discrete=True, depth=2),
              simple=dict(simple_ray_casting=True))[mode]
    done = 0
    for it in range(3):
        o, xyz, _ = scans.random_cloud(400, seed=seed * 10 + it, extent=30.0, origin=(0.3 + 9 * it, -0.2, 0.4))
        mr = [-1.0, 12.0, 40.0][it]
        try:
            b.insert(o, xyz, max_range=mr, **kw)  # the port goes first: it refuses runaway rays
        except RunawayRay:
            continue  # the reference would walk ~2^31 cells on this input (see ufo_oracle.cpp)
        a.insert(o, xyz, max_range=mr, **kw)
        done += 1
    assert done >= 2
    _assert_same(a, b)


@pytest.mark.parametrize("discrete", [False, True])
def test_full_lidar_scan(discrete, ref_available, port_available):
    """BASELINE configs C1 (continuous) / C2 (discrete): 131 072 points, 16 cm, 20 m."""
    if not ref_available:
        pytest.skip("reference build not available")
    a, b = _pair(resolution=0.16)
    o, xyz, _ = scans.lidar64()
    a.insert(o, xyz, max_range=20.0, discrete=discrete)
    b.insert(o, xyz, max_range=20.0, discrete=discrete)
    _assert_same(a, b)


def test_colour_sequence(ref_available, port_available):
    if not ref_available:
        pytest.skip("reference build not available")
    a, b = _pair(resolution=0.08, color=True)
    for s in range(3):
        o, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(s + 2), seed=100 + s, beams=16, azimuths=256, colored=True)
        a.insert(o, xyz, rgb, max_range=12.0, discrete=True)
        b.insert(o, xyz, rgb, max_range=12.0, discrete=True)
    _assert_same(a, b)


def test_pruning_collapse_sequence(ref_available, port_available):
    """Ten identical scans: free space saturates at clamp_min and collapses (history-dependent pruning)."""
    if not ref_available:
        pytest.skip("reference build not available")
    a, b = _pair(resolution=0.16)
    o, xyz, _ = scans.lidar64(beams=32, azimuths=512)
    for _ in range(10):
        a.insert(o, xyz, max_range=20.0, discrete=True)
        b.insert(o, xyz, max_range=20.0, discrete=True)
    _assert_same(a, b)


def _pc2_records(n=4000, step=32, seed=5, nan_every=13):
    """A PointCloud2-style record stream: float32 x, y, z at 0/4/8, packed rgb (b, g, r, a) at 16, padding."""
    rng = np.random.default_rng(seed)
    buf = rng.integers(0, 256, (n, step), dtype=np.uint8)  # padding bytes are arbitrary
    xyz = rng.uniform(-9.0, 9.0, (n, 3)).astype(np.float32)
    xyz[::nan_every, rng.integers(0, 3)] = np.nan
    buf[:, 0:12] = xyz.view(np.uint8).reshape(n, 12)
    return buf, (0, 4, 8), (18, 17, 16)


def test_ingest_port_equals_reference_pose6_transform():
    """rosToUfo + PointCloud::transform: the restatement's quaternion arithmetic equals the reference's own
    Pose6::transform bit for bit, NaN points dropped, colour bytes taken along."""
    import oracle
    buf, oxyz, orgb = _pc2_records()
    rng = np.random.default_rng(9)
    for k in range(5):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.uniform(-3, 3, 3)
        a = oracle.ingest(buf, buf.shape[1], oxyz, orgb, q, t, "port")
        b = oracle.ingest(buf, buf.shape[1], oxyz, orgb, q, t, "reference")
        assert a[0].shape[0] == buf.shape[0] - len(range(0, buf.shape[0], 13))
        assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64))
        assert np.array_equal(a[1], b[1])
    # identity pose: float32 -> float64 conversion only (plus the +-0 arithmetic of the quaternion product)
    a = oracle.ingest(buf, buf.shape[1], oxyz, None, [1.0, 0, 0, 0], [0.0, 0, 0], "port")
    keep = ~np.isnan(buf[:, 0:12].copy().view(np.float32).reshape(-1, 3)).any(axis=1)
    assert np.array_equal(a[0], buf[:, 0:12].copy().view(np.float32).reshape(-1, 3)[keep].astype(np.float64))
    assert not a[1].any()


@pytest.mark.parametrize("color", [False, True])
def test_set_value_volume_port_equals_reference(color):
    """Robot clearing: setValueVolume(AABB, getClampingThresMin(), min_depth) after every
    scan, as the server does; min_depth 0, 1, 2 (subtrees deleted), pruning on and off."""
    from ufomap_amd import scans
    from oracle import OracleMap
    for pruning in (True, False):
        p = OracleMap(0.16, color=color, automatic_pruning=pruning, kind="port")
        r = OracleMap(0.16, color=color, automatic_pruning=pruning, kind="reference")
        assert p.clamping_thres() == r.clamping_thres()
        for s in range(3):
            origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=scans.lidar_pose(s), seed=3 + s, colored=color)
            for m in (p, r):
                m.insert(origin, xyz, rgb if color else None, max_range=10.0, discrete=True)
            c = np.array(origin)
            for md in (0, 1, 2, 0):
                ext = np.array([0.45, 0.45, 0.6]) * (1 + md)
                for m in (p, r):
                    m.setValueVolume(c - ext, c + ext, m.clamping_thres()[0], md)
                assert same_dump(p.leaves(True), r.leaves(True)), (pruning, s, md)
                assert same_dump(p.inner(), r.inner()), (pruning, s, md)
                assert p.write() == r.write()
    # a volume outside the map, and min_depth beyond the tree: no-ops
    before = p.write()
    p.setValueVolume([1e7, 1e7, 1e7], [2e7, 2e7, 2e7], 0.2, 0)
    p.setValueVolume([-1, -1, -1], [1, 1, 1], 0.2, 40)
    assert p.write() == before


@pytest.mark.parametrize("color", [False, True])
def test_point_queries_port_equals_reference(color):
    """getState / contains* / the node's log-odds through the reference's own getNode (with
    its depth convention) for points near surfaces, in free and unknown space and kilometres away."""
    from ufomap_amd import scans
    from oracle import OracleMap
    p = OracleMap(0.16, color=color, kind="port")
    r = OracleMap(0.16, color=color, kind="reference")
    for s in range(3):
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=scans.lidar_pose(s), seed=3 + s, colored=color)
        for m in (p, r):
            m.insert(origin, xyz, rgb if color else None, max_range=10.0, discrete=True)
    rng = np.random.default_rng(0)
    q = np.concatenate([xyz[::7] + rng.normal(0, 0.05, xyz[::7].shape), rng.uniform(-15, 15, (3000, 3)), rng.uniform(-4000, 4000, (500, 3))])
    for depth in (0, 1, 2, 5, 14, 15):
        a, b = p.query(q, depth), r.query(q, depth)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]), depth
    assert len(set(p.query(q, 0)[1] & 7)) == 3  # occupied, free and unknown all occur


@pytest.mark.parametrize("early", [1, 3, 10])
@pytest.mark.parametrize("mode", ["continuous", "discrete", "simple", "discrete_d1"])
def test_early_stopping(early, mode, ref_available, port_available):
    """`early_stopping` (occupancy_map_base.h:1289-1298, 1327-1333; the server's dynamic-reconfigure range is 0..10): a ray ends once
    that many cells in a row were in the scan's set already -- order-dependent, pinned against the reference on a LiDAR sweep
    (neighbouring rays share most cells) and on random clouds with clipping."""
    if not ref_available:
        pytest.skip("reference build not available")
    kw = dict(continuous=dict(), discrete=dict(discrete=True), simple=dict(discrete=True, simple_ray_casting=True),
              discrete_d1=dict(discrete=True, depth=1))[mode]
    a, b = _pair(resolution=0.16)
    for s in range(3):
        o, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=scans.lidar_pose(s), seed=100 + s)
        a.insert(o, xyz, max_range=12.0, early_stopping=early, **kw)
        b.insert(o, xyz, max_range=12.0, early_stopping=early, **kw)
    _assert_same(a, b)
    assert b.last_steps() < OracleMap(0.16, kind="port").last_steps() + 10 ** 9  # (the port reports its step count with early stopping too)
    a, b = _pair(resolution=0.5, depth_levels=6)
    for it in range(2):
        o, xyz, _ = scans.random_cloud(300, seed=77 + it, extent=30.0, origin=(0.3 + 9 * it, -0.2, 0.4))
        try:
            b.insert(o, xyz, max_range=[-1.0, 40.0][it], early_stopping=early, **kw)
        except RunawayRay:
            continue
        a.insert(o, xyz, max_range=[-1.0, 40.0][it], early_stopping=early, **kw)
    _assert_same(a, b)
