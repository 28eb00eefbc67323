"""SURVEY.md 8 rows finished in round 2 (run with -m gpu): the per-code change set (a18 / f4), device leaf / tree
iteration with bounding volume + state filter (f3), write / writeData with bounding volume, min depth and LZ4 and read /
readData into the GPU map (f1), the sensor-model accessors and setOccupiedFreeThres. The checker is the UNMODIFIED
reference build (oracle/_ref/libufo_ref.so travels to the GPU box); the port does not restate these rows."""
import itertools

import numpy as np
import pytest

from conftest import same_dump

pytestmark = pytest.mark.gpu


def _maps(color=False, **params):
    import oracle
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    if not oracle.available("reference"):
        pytest.skip("oracle/_ref/libufo_ref.so not present")
    g = (OccupancyMapColor if color else OccupancyMap)(**params)
    o = OracleMap(kind="reference", color=color, **params)
    return g, o


def _insert(g, o, origin, xyz, rgb=None, **kw):
    from ufomap_amd import PointCloud, PointCloudColor
    cloud = PointCloudColor(xyz, rgb) if rgb is not None else PointCloud(xyz)
    fn = g.insertPointCloudDiscrete if kw.get("discrete") else g.insertPointCloud
    fn(origin, cloud, kw.get("max_range", -1.0), kw.get("depth", 0))
    o.insert(origin, xyz, rgb, **kw)


def _populate(g, o, color=False, n=3, **kw):
    from ufomap_amd import scans
    for s in range(n):
        origin, xyz, rgb = scans.lidar64(beams=24, azimuths=384, origin=scans.lidar_pose(s), seed=40 + s, colored=color)
        _insert(g, o, origin, xyz, rgb if color else None, max_range=9.0, discrete=True, **kw)
    return np.array(origin)


def _same(a, b, what):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and np.array_equal(x, y), f"{what}: output {k} differs ({x.shape} vs {y.shape})"


@pytest.mark.parametrize("color", [False, True])
def test_leaf_and_tree_iteration_with_bounding_volume_and_state_filter(color):
    """beginLeaves / beginTree run to the end: every combination of bounding volume, the three state switches, `contains`
    and min_depth that changes the predicates of iterator/occupancy_map.h:168-207 -- same nodes, same order, same values."""
    g, o = _maps(color=color, resolution=0.16)
    c = _populate(g, o, color)
    g.setValueVolume(c - 0.6, c + 0.6, g.getClampingThresMin(), 1)
    o.setValueVolume(c - 0.6, c + 0.6, o.clamping_thres()[0], 1)
    boxes = [None, (c + [1.0, 0.5, -0.3], [2.1, 1.7, 0.9]), ([0.08, 0.08, 0.08], [0.01, 0.01, 0.01]), ([900.0, 0, 0], [1, 1, 1])]
    n_checked = 0
    for aabb, (occ, fre, unk), contains, min_depth, only_leaves in itertools.product(
            boxes, [(1, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)], [False, True], [0, 1, 3, 15], [True, False]):
        if aabb is None and unk and min_depth == 0 and only_leaves is False and not contains:
            continue  # (nothing wrong with it: just the one combination that returns the whole tree four times over)
        a = g.iterate(aabb, occ, fre, unk, contains, min_depth, only_leaves)
        b = o.iterate(aabb, occ, fre, unk, contains, min_depth, only_leaves)
        _same(a, b, f"aabb={aabb is not None} occ={occ} free={fre} unk={unk} contains={contains} min_depth={min_depth} leaves={only_leaves}")
        n_checked += len(a[0])
    assert n_checked > 100000
    # fresh map: the root alone
    g2, o2 = _maps(color=color, resolution=0.16)
    for args in [(None, 1, 1, 1, False, 0, True), (None, 1, 1, 0, False, 0, True), (None, 1, 1, 1, True, 3, False)]:
        _same(g2.iterate(*args), o2.iterate(*args), f"fresh map {args}")


def test_change_detection_code_set_deep_trees():
    """depth_levels = 20 / 21: a depth-0 code is 60 / 63 bits wide -- the change log's records (table.h: ChangeLog) must hold it
    together with the depth (a depth field in the top bits, as the records used to be packed, collides from 20 levels on).
    20 levels: against the reference. 21 levels: the reference smashes its stack there (insertPointCloud*), so the same
    scans must give change sets of the same sizes and depths as at 20 levels (same resolution: the same voxels)."""
    from ufomap_amd import OccupancyMap, scans
    g, o = _maps(resolution=0.16, depth_levels=20)
    g21 = OccupancyMap(0.16, depth_levels=21)
    for m in (g, o, g21):
        m.enableChangeDetection(True)
    for s in range(4):
        # positive AND negative coordinates: the high Morton bits of a code are set for non-negative coordinates
        origin, xyz, _ = scans.lidar64(beams=8, azimuths=128, origin=(0.1 + 40.0 * (s & 1), -0.2, 1.7), seed=21 + s)
        depth = 1 if s == 2 else 0
        kw = dict(max_range=8.0, depth=depth, discrete=bool(s & 1) or depth > 0)
        _insert(g, o, origin, xyz, None, **kw)
        _insert(g21, _Null(), origin, xyz, None, **kw)
        _same(g.changes(), o.changes(), f"20 levels, scan {s}")
        c20, c21 = g.changes(), g21.changes()
        assert len(c20[0]) == len(c21[0]) and np.array_equal(np.sort(c20[1]), np.sort(c21[1])), f"21 levels, scan {s}"
    assert len(g.changes()[0]) > 500 and int(g.changes()[0].max()) >> 58 != 0, "no code reached the bits the old packing used for the depth"
    assert int(g21.changes()[0].max()) >> 61 != 0
    assert same_dump(g.leaves(True), o.leaves(True))


class _Null:
    def insert(self, *a, **k):
        pass


@pytest.mark.parametrize("color", [False, True])
def test_change_detection_code_set(color):
    """enableChangeDetection: every leaf update that changes a value records its code (occupancy_map_base.h:1070-1072,
    1094-1108; occupancy_map_color.h:278-280) -- depth-0 inserts (hits and misses, saturation: later scans change less),
    coarse misses over expanded nodes (insert depth 2), reset in between."""
    from ufomap_amd import scans
    g, o = _maps(color=color, resolution=0.16)
    g.enableChangeDetection(True)
    o.enableChangeDetection(True)
    sizes = []
    for s in range(9):
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=scans.lidar_pose(s % 2), seed=7 + (s % 2), colored=color)
        depth = 2 if s in (3, 6) else 0
        _insert(g, o, origin, xyz, rgb if color else None, max_range=8.0, depth=depth, discrete=True if color else bool(s & 1) or depth > 0)
        _same(g.changes(), o.changes(), f"scan {s}")
        sizes.append(len(g.changes()[0]))
        if s == 4:
            g.resetChangeDetection()
            o.resetChangeDetection()
            assert len(g.changes()[0]) == 0
    assert sizes[0] > 1000 and sizes[5] < sizes[4]
    assert same_dump(g.leaves(True), o.leaves(True))
    if not color:  # (the tiled tree update logs the changed voxels itself: k_tile)
        assert g.debug()[61] >= 3, f"change detection kept the scans off the fast path: {g.debug()[58:64]}"
    g.enableChangeDetection(False)
    o.enableChangeDetection(False)
    before = len(g.changes()[0])
    _insert(g, o, origin, xyz, rgb if color else None, max_range=8.0, discrete=True)
    assert len(g.changes()[0]) == before == len(o.changes()[0])


@pytest.mark.parametrize("color", [False, True])
def test_write_with_bounding_volume_min_depth_and_lz4(color):
    """Octree::write / writeData with every argument: what the server's SaveMap (server.cpp:389) and every published
    UFOMap message (ufoToMsg -> writeData, conversions.h:162-186; server.cpp:200) contain -- byte for byte, LZ4 included."""
    g, o = _maps(color=color, resolution=0.16)
    c = _populate(g, o, color)
    boxes = [None, (c + [1.0, 0.5, -0.3], [2.1, 1.7, 0.9]), ([0.08, 0.08, 0.08], [0.01, 0.01, 0.01]), ([900.0, 0, 0], [1, 1, 1])]
    for aabb, min_depth, header in itertools.product(boxes, [0, 1, 2, 5, 16, 17], [True, False]):
        a, b = g.write_ex(aabb, False, min_depth, header=header), o.write_ex(aabb, False, min_depth, header=header)
        assert a[0] == b[0], f"aabb={aabb is not None} min_depth={min_depth} header={header}: bytes differ ({len(a[0])} vs {len(b[0])})"
        assert header or a[1] == b[1]
    for aabb, (accel, level) in itertools.product(boxes[:2], [(1, 0), (8, 0), (1, 4)]):
        a, b = g.write_ex(aabb, True, 0, accel, level, header=False), o.write_ex(aabb, True, 0, accel, level, header=False)
        assert a == b, f"compressed aabb={aabb is not None} accel={accel} level={level}"
    assert g.write_ex(None, True, 1)[0] == o.write_ex(None, True, 1)[0]
    # the same stream the long way (what maps beyond a million blocks take: the host sizes the list and the output between the passes)
    g.set_option("ser_short", 0)
    for aabb, min_depth in itertools.product(boxes[:3], [0, 2]):
        assert g.write_ex(aabb, False, min_depth, header=False) == o.write_ex(aabb, False, min_depth, header=False), f"long way: aabb={aabb is not None} min_depth={min_depth}"


@pytest.mark.parametrize("color", [False, True])
def test_read_and_read_data_into_the_gpu_map(color):
    """Octree::read / readData (octree.h:701-777) -> readNodes (occupancy_map_base.h:1379-1455): a whole file into a fresh
    map, an LZ4 message into a fresh map, a partial (bounding-volume, min_depth) message merged into a DIFFERENT populated
    map (subtrees replaced, leaves expanded, ancestors re-evaluated and pruned), a root-only stream."""
    from ufomap_amd import scans
    g, o = _maps(color=color, resolution=0.16)
    c = _populate(g, o, color)
    whole, _ = o.write_ex()
    lz, lz_size = o.write_ex(compress=True, header=False)
    box = (c + [1.0, 0.5, -0.3], [2.1, 1.7, 0.9])
    part, part_size = o.write_ex(box, False, 1, header=False)

    def check(ga, oa, what):
        assert same_dump(ga.leaves(True), oa.leaves(True)), what + ": leaves differ"
        assert same_dump(ga.inner(), oa.inner()), what + ": inner nodes differ"
        assert ga.write() == oa.write_ex()[0], what + ": byte stream differs"

    g1, o1 = _maps(color=color, resolution=0.16)
    g1.read(whole)
    o1.read(whole)
    check(g1, o1, "read(whole file)")
    check(g1, o, "read(whole file) vs the map it came from")
    g2, o2 = _maps(color=color, resolution=0.08, depth_levels=14)  # other geometry: readData clears to the stream's
    g2.readData(lz, 0.16, 16, lz_size, True)
    o2.readData(lz, 0.16, 16, lz_size, True)
    check(g2, o2, "readData(LZ4)")
    g3, o3 = _maps(color=color, resolution=0.16)
    for s in range(2):
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=tuple(c + [0.7, -0.4, 0.1 * s]), seed=90 + s, colored=color)
        _insert(g3, o3, origin, xyz, rgb if color else None, max_range=6.0, discrete=True)
    g3.readData(part, 0.16, 16, part_size, False, box)
    o3.readData(part, 0.16, 16, part_size, False, box)
    check(g3, o3, "readData(partial, merged)")
    g3.readData(whole[whole.index(b"data\n") + 5:], 0.16, 16)
    o3.readData(whole[whole.index(b"data\n") + 5:], 0.16, 16)
    check(g3, o3, "readData(whole, merged)")
    g4, o4 = _maps(color=color, resolution=0.16)
    root_only, n = o4.write_ex(header=False)
    g3.readData(root_only, 0.16, 16, n)
    o3.readData(root_only, 0.16, 16, n)
    check(g3, o3, "readData(root only)")


def test_sensor_model_accessors_and_threshold_change():
    """Getters (toProb of the float-narrowed logits), the four setters the server's reconfigure callback uses
    (server.cpp:468-471) and setOccupiedFreeThres on a populated map (the reference rewrites the tree through
    write + read so that every contains_free / contains_unknown follows the new thresholds)."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16, prob_hit=0.8, prob_miss=0.35, clamping_thres_min=0.2, clamping_thres_max=0.9)
    assert g.sensor_model() == o.sensor_model()
    _populate(g, o)
    for which, fn, val in [(2, g.setProbHit, 0.75), (3, g.setProbMiss, 0.45), (4, g.setClampingThresMin, 0.1), (5, g.setClampingThresMax, 0.95)]:
        fn(val)
        o.set_model_value(which, val)
    assert g.sensor_model() == o.sensor_model()
    assert g.getClampingThresMin() == o.clamping_thres()[0]
    _populate(g, o, n=2)
    assert same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner())
    g.setOccupiedFreeThres(0.7, 0.3)
    o.setOccupiedFreeThres(0.7, 0.3)
    assert g.sensor_model() == o.sensor_model()
    assert same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.leaves(False), o.leaves(False))
    assert same_dump(g.inner(), o.inner()), "inner flags after setOccupiedFreeThres differ"
    _populate(g, o, n=2)
    assert same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner())
    q = np.random.default_rng(0).uniform(-8, 8, (3000, 3))
    a, b = g.query(q, 1), o.query(q, 1)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])


def test_minmax_change_detection_switch_and_clear_to():
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    origin, xyz, _ = scans.lidar64(beams=8, azimuths=128)
    for m in (g, o):
        m.enableMinMaxChangeDetection(False)
    _insert(g, o, origin, xyz, max_range=8.0, discrete=True)
    assert same_dump(g.minmax_change(), o.minmax_change())
    for m in (g, o):
        m.enableMinMaxChangeDetection(True)
    _insert(g, o, origin + 0.3, xyz, max_range=8.0, discrete=True)
    assert same_dump(g.minmax_change(), o.minmax_change())
    g.clear_to(0.05, 12)
    o.clear_to(0.05, 12)
    _insert(g, o, origin, xyz, max_range=3.0, discrete=True)
    assert same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner())
    assert g.write() == o.write_ex()[0]


def test_device_expf_equals_the_hosts_libm():
    """toProb's std::exp(float) on the device (expf_ref.h) against this box's libm -- the function the reference calls -- on
    16 M random arguments in [-16, 16] and the special values (the host-side sweep of EVERY float in that range runs in the
    CPU suite: tests/test_toprob_sweep.py; here the device is shown to run the same function)."""
    import ctypes as C
    from test_toprob_sweep import _build
    from ufomap_amd import capi
    lib, host = capi.load(), C.CDLL(_build(shared=True))
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-16, 16, 1 << 24).astype(np.float32), (rng.standard_normal(1 << 20) * 1e-3).astype(np.float32),
                        np.array([0.0, -0.0, 88.7, 88.73, 100.0, -103.9, -104.0, -200.0, np.inf, -np.inf, 1e-45, -1e-45], np.float32)])
    dev, ref = np.empty_like(x), np.empty_like(x)
    capi.check(lib.ufomap_dev_expf(C.c_void_p(x.ctypes.data), C.c_void_p(dev.ctypes.data), C.c_size_t(x.size), 0))
    host.expf_libm(C.c_void_p(x.ctypes.data), C.c_void_p(ref.ctypes.data), C.c_size_t(x.size))
    bad = np.nonzero(dev.view(np.uint32) != ref.view(np.uint32))[0]
    assert bad.size == 0, f"{bad.size} values differ, e.g. exp({x[bad[0]]!r}) = {dev[bad[0]]!r} on the device, {ref[bad[0]]!r} on the host"
