"""HIP path vs oracle -- the parity tests proper (run with -m gpu on an MI355X).

Everything goes through the C ABI (ufomap_amd.occupancy_map -> csrc/libufomap_hip.so).  Bar:
bit-exact Morton codes / depths / colours and bit-exact float32 log-odds (north_star allows 1e-5;
we hold 0), including the leaf structure (pruning), inner-node summaries and the change AABB.
"""
import numpy as np
import pytest

import golden_util
from conftest import same_dump

pytestmark = pytest.mark.gpu


def _maps(color=False, **params):
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    g = (OccupancyMapColor if color else OccupancyMap)(**params)
    o = OracleMap(kind="port", color=color, **params)
    return g, o


def _gpu_insert(g, origin, xyz, rgb=None, max_range=-1.0, depth=0, discrete=False, simple_ray_casting=False, async_=False):
    from ufomap_amd import PointCloud, PointCloudColor
    cloud = PointCloudColor(xyz, rgb) if rgb is not None else PointCloud(xyz)
    fn = g.insertPointCloudDiscrete if discrete else g.insertPointCloud
    fn(origin, cloud, max_range, depth, simple_ray_casting, 0, async_)


def _assert_same_map(g, o, what=""):
    gl, ol = g.leaves(True), o.leaves(True)
    assert len(gl[0]) == len(ol[0]), f"{what}: leaf count {len(gl[0])} vs oracle {len(ol[0])}"
    assert np.array_equal(gl[0], ol[0]) and np.array_equal(gl[1], ol[1]), f"{what}: leaf codes/depths differ"
    assert np.array_equal(gl[2], ol[2]), f"{what}: log-odds differ, max |d| = {np.abs(gl[2] - ol[2]).max()}"
    assert np.array_equal(gl[3], ol[3]), f"{what}: colours differ"
    assert same_dump(g.inner(), o.inner()), f"{what}: inner-node dump differs"
    assert same_dump(g.minmax_change(), o.minmax_change()), f"{what}: change AABB differs"
    assert g.write() == o.write(), f"{what}: map byte stream differs from the reference format writer"


@pytest.mark.parametrize("name", golden_util.names())
def test_gpu_matches_golden(name):
    """Golden vectors generated from the unmodified reference (tests/golden/make_golden.py)."""
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    g = golden_util.Golden(name)
    params = dict(g.params)
    color = params.pop("color", False)
    m = (OccupancyMapColor if color else OccupancyMap)(**params)
    for origin, xyz, rgb, kw in g.scans():
        _gpu_insert(m, origin, xyz, rgb, **kw)
    g.check(m)


@pytest.mark.parametrize("discrete", [False, True])
def test_stage_level_hits_and_misses(discrete):
    """Unique hit codes and unique miss codes of one scan, bit-exact (SURVEY 4 iii)."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    origin, xyz, _ = scans.lidar64(beams=32, azimuths=1024)
    _gpu_insert(g, origin, xyz, max_range=20.0, discrete=discrete)
    o.insert(origin, xyz, max_range=20.0, discrete=discrete)
    assert np.array_equal(g.last_hits(), o.last_hits())
    assert np.array_equal(g.last_misses(), o.last_misses())
    c = g.last_counts()
    assert c["rays"] == len(o.last_rays())
    assert c["steps"] == o.last_steps()
    _assert_same_map(g, o, "stage")


@pytest.mark.parametrize("discrete", [False, True])
def test_full_lidar_scan_c1_c2(discrete):
    """BASELINE configs C1 (continuous) and C2 (discrete): 131 072 points, 16 cm, 20 m."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    origin, xyz, _ = scans.lidar64()
    _gpu_insert(g, origin, xyz, max_range=20.0, discrete=discrete)
    o.insert(origin, xyz, max_range=20.0, discrete=discrete)
    _assert_same_map(g, o, "C1/C2")


def test_batch_of_8_poses_c4_sequential():
    """BASELINE config C4 integrated sequentially into one map (8 poses)."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    for s in range(8):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s)
        _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
    _assert_same_map(g, o, "C4")


@pytest.mark.parametrize("merge", [0, 1])
def test_one_pass_and_two_pass_map_update_agree(merge):
    """Depth-0 scans update the tree in one pass (hits + misses merged) by default; the two-pass form
    (also used for insert depth > 0 and for update lists from other GPUs) must give the same map:
    moving sensor, colour, overlapping hit/miss voxels, saturation and collapse."""
    from ufomap_amd import scans
    for color in (True, False):
        g, o = _maps(color=color, resolution=0.16)
        g.set_option("merge_phases", merge)
        for s in range(6):
            origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=scans.lidar_pose(s % 3), seed=7 + (s % 3), colored=color)
            discrete = True if color else bool(s & 1)  # the colour overload exists for the discrete integrator only
            _gpu_insert(g, origin, xyz, rgb, max_range=12.0, discrete=discrete)
            o.insert(origin, xyz, rgb, max_range=12.0, discrete=discrete)
            _assert_same_map(g, o, f"merge={merge} color={color} scan {s}")


def test_repeated_scan_saturation_and_pruning():
    """Ten identical scans: clamping at both ends and history-dependent collapse of free space."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    origin, xyz, _ = scans.lidar64(beams=32, azimuths=512)
    for i in range(10):
        _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
        _assert_same_map(g, o, f"scan {i}")


def test_colour_c5_8cm():
    """BASELINE config C5: coloured scan, 8 cm leaf, occupancy + RGB fused update."""
    from ufomap_amd import scans
    g, o = _maps(color=True, resolution=0.08)
    for s in range(2):
        origin, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(s + 3), seed=100 + s, colored=True)
        _gpu_insert(g, origin, xyz, rgb, max_range=20.0, discrete=True)
        o.insert(origin, xyz, rgb, max_range=20.0, discrete=True)
    _assert_same_map(g, o, "C5")


@pytest.mark.parametrize("depth", [3, 6])
def test_rgbd_c3_insert_depth(depth):
    """BASELINE config C3 (2 mm, 5 m, 307 200 points) with free space cleared at depth 3 / 6."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.002)
    origin, xyz, _ = scans.rgbd()
    for _ in range(2):
        _gpu_insert(g, origin, xyz, max_range=5.0, depth=depth, discrete=True)
        o.insert(origin, xyz, max_range=5.0, depth=depth, discrete=True)
    _assert_same_map(g, o, f"C3 depth {depth}")


def test_async_and_wait():
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    origin, xyz, _ = scans.lidar64(beams=16, azimuths=512)
    _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True, async_=True)
    g.insertPointCloudWait()
    assert g.insertPointCloudDone()
    _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True, async_=True)  # joins nothing pending, then enqueues
    o.insert(origin, xyz, max_range=20.0, discrete=True)
    o.insert(origin, xyz, max_range=20.0, discrete=True)
    _assert_same_map(g, o, "async")


def test_edge_cases_empty_and_errors():
    from ufomap_amd import OccupancyMap, capi
    with pytest.raises(ValueError):
        OccupancyMap(0.1, depth_levels=1)
    with pytest.raises(ValueError):
        OccupancyMap(0.1, depth_levels=22)
    m = OccupancyMap(0.16)
    _gpu_insert(m, [0, 0, 0], np.zeros((0, 3)), max_range=20.0, discrete=True)
    assert len(m.leaves()[0]) == 0
    codes, depths, occ, _ = m.leaves(True)
    assert codes.tolist() == [0] and depths.tolist() == [16] and occ.tolist() == [0.0]
    # (until round 5 early_stopping needed a dense first-ray array over the ray box -- 4e6 cells do not fit 1 MiB: UNSUPPORTED. Now the
    # cells the rays visit go into a sparse set: the call succeeds and equals the reference)
    m.set_scratch_limit(1 << 20)
    pts = np.array([[20.0, 20.0, 20.0], [1.0, 1.0, 1.0], [19.9, 20.0, 20.0]])
    m.insertPointCloud([0, 0, 0], pts, -1.0, 0, False, 3)
    _, o = _maps(resolution=0.16)
    o.insert([0, 0, 0], pts, max_range=-1.0, early_stopping=3)
    assert same_dump(m.leaves(True), o.leaves(True)) and same_dump(m.inner(), o.inner())


def test_runaway_ray_is_refused_and_map_unchanged():
    """The input on which the reference walks ~2^31 cells: refused, like the oracle port does."""
    from oracle import RunawayRay
    from ufomap_amd import capi
    g, o = _maps(resolution=0.5, depth_levels=6)
    origin = np.array([float.fromhex("0x1.24ccccccccccdp+4"), float.fromhex("-0x1.999999999999ap-3"), float.fromhex("0x1.999999999999ap-2")])
    pt = np.array([[float.fromhex("0x1.cbec1074f2bf0p+2"), float.fromhex("-0x1.d4ce8bc744bffp+4"), float.fromhex("-0x1.7e4cd8d9a9637p+3")]])
    with pytest.raises(RunawayRay):
        o.insert(origin, pt, max_range=40.0, discrete=True)
    with pytest.raises(capi.UfomapError) as e:
        _gpu_insert(g, origin, pt, max_range=40.0, discrete=True)
    assert e.value.code in (capi.ERR_RUNAWAY, capi.ERR_CAPACITY)
    assert len(g.leaves()[0]) == 0


@pytest.mark.parametrize("merged_lists", [1, 0])
def test_split_path_equals_sequential_c4(merged_lists):
    """scan_keys + apply_keys (the multi-GPU split of the path) applied in scan order on one GPU equals
    sequential insertPointCloudDiscrete of the same 8 scans (BASELINE config C4), bit for bit -- with the
    merged per-scan list (one record per block) and with separate hit and miss lists."""
    import torch
    from ufomap_amd import OccupancyMap, scans
    from ufomap_amd.dist import ENTRY_BYTES
    seq, o = _maps(resolution=0.16)
    split = OccupancyMap(0.16)
    scanner = OccupancyMap(0.16)  # plays "another GPU": only ever scans, its own map stays empty
    scanner.set_option("merge_phases", merged_lists)
    lists = []
    for s in range(8):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, beams=32, azimuths=1024)
        _gpu_insert(seq, origin, xyz, max_range=20.0, discrete=True)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
        d = torch.from_numpy(xyz).cuda()
        info = scanner.scan_keys(origin, d.data_ptr(), xyz.shape[0], 20.0, 0, True)
        buf = torch.empty((info.n_hit + info.n_miss) * ENTRY_BYTES, dtype=torch.uint8, device="cuda")
        scanner.get_keys(buf.data_ptr(), info.n_hit + info.n_miss, info)
        lists.append((info, buf))
    assert len(scanner.leaves()[0]) == 0  # scanning never touches the map
    for info, buf in lists:
        split.apply_keys(buf.data_ptr(), info)
    _assert_same_map(seq, o, "sequential")
    assert same_dump(split.leaves(True), o.leaves(True)), "split path: leaves differ"
    assert same_dump(split.inner(), o.inner()), "split path: inner nodes differ"


@pytest.mark.parametrize("chunk", [8, 3])
def test_batch_apply_equals_sequential_c4(chunk):
    """BASELINE config C4: the update lists of a batch of scans applied with ONE walk of the tree
    (ufomap_map_apply_keys_batch) give the map the reference builds by integrating the scans one after
    the other -- values, inner nodes and the history-dependent leaf structure; two rounds over the same
    poses so that clamping and collapse are exercised across batches."""
    import torch
    from ufomap_amd import scans, OccupancyMap
    g, o = _maps(resolution=0.16)
    scanner = OccupancyMap(0.16)  # never integrates: only casts rays
    bufs, infos = [], []
    for rnd in range(2):
        for s0 in range(0, 8, chunk):
            bufs.clear(); infos.clear()
            for s in range(s0, min(8, s0 + chunk)):
                origin, xyz, _ = scans.lidar64(beams=32, azimuths=1024, origin=scans.lidar_pose(s), seed=100 + s)
                d = torch.from_numpy(xyz).cuda()
                info = scanner.scan_keys(origin, d.data_ptr(), xyz.shape[0], 20.0, 0, True)
                buf = torch.empty((info.n_hit + info.n_miss) * 16, dtype=torch.uint8, device="cuda")
                scanner.get_keys(buf.data_ptr(), buf.numel() // 16, info)
                bufs.append(buf); infos.append(info)
                o.insert(origin, xyz, max_range=20.0, discrete=True)
            g.apply_keys_batch([b.data_ptr() for b in bufs], infos)
            what = f"batch round {rnd} chunk from {s0}"  # (the change AABB belongs to the scanning side)
            assert same_dump(g.leaves(True), o.leaves(True)), what + ": leaves differ"
            assert same_dump(g.inner(), o.inner()), what + ": inner nodes differ"


@pytest.mark.parametrize("color", [False, True])
def test_pointcloud2_ingest_fused_equals_reference_chain(color):
    """raw PointCloud2 records -> NaN filter -> Pose6 transform -> insertPointCloudDiscrete,
    fused into the scan's first kernel, against the oracle's restatement of the reference chain
    (rosToUfo, PointCloud::transform, insertPointCloudDiscrete): same map bit for bit, host and device data."""
    import torch
    import oracle
    from ufomap_amd import scans
    g, o = _maps(color=color, resolution=0.16)
    rng = np.random.default_rng(3)
    for s in range(3):
        # a LiDAR scan in the sensor frame, as float32 records with NaN returns and a moving pose
        _, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=(0.0, 0.0, 0.0), seed=20 + s, colored=True)
        n = xyz.shape[0]
        step = 32
        buf = rng.integers(0, 256, (n, step), dtype=np.uint8)
        f = xyz.astype(np.float32)
        f[::29, s % 3] = np.nan
        buf[:, 0:12] = f.view(np.uint8).reshape(n, 12)
        buf[:, 16], buf[:, 17], buf[:, 18] = rgb[:, 2], rgb[:, 1], rgb[:, 0]  # packed "rgb": b, g, r
        q = np.array([np.cos(0.1 * (s + 1)), 0.02 * s, -0.03, np.sin(0.1 * (s + 1))])
        q /= np.linalg.norm(q)
        t = np.array([0.4 * s, -0.2 * s, 0.9])
        oxyz, orgb = (0, 4, 8), (18, 17, 16)
        if s == 1:
            d = torch.from_numpy(buf.reshape(-1)).cuda()
            g.insertPointCloud2(t, q, d.data_ptr(), step, oxyz, orgb if color else None, max_range=15.0, n_points=n)
        else:
            g.insertPointCloud2(t, q, buf, step, oxyz, orgb if color else None, max_range=15.0)
        cxyz, crgb = oracle.ingest(buf, step, oxyz, orgb if color else None, q, t, "port")
        assert cxyz.shape[0] < n
        o.insert(t, cxyz, crgb if color else None, max_range=15.0, discrete=True)
        _assert_same_map(g, o, f"pointcloud2 scan {s}")
    # the same records again from the last pose: on the ray grid predicted from the scans before, i.e. through the tiled tree
    # update -- colours included (they reach it in the set's own array, written by the first kernel that loads the points)
    before = g.debug()[61]
    for _ in range(2):
        g.insertPointCloud2(t, q, buf, step, oxyz, orgb if color else None, max_range=15.0)
        o.insert(t, cxyz, crgb if color else None, max_range=15.0, discrete=True)
    _assert_same_map(g, o, "pointcloud2, repeated scans")
    assert g.debug()[61] > before, "PointCloud2 records did not take the fast path"


@pytest.mark.parametrize("color", [False, True])
def test_set_value_volume_robot_clearing(color):
    """the server's per-scan robot clearing -- setValueVolume(AABB around the sensor,
    getClampingThresMin(), clearing depth) after every insert -- level-synchronous on the GPU against the
    reference's recursion: values, inner nodes, pruned structure, byte stream; min_depth 0, 1, 2."""
    from ufomap_amd import scans
    g, o = _maps(color=color, resolution=0.16)
    assert g.getClampingThresMin() == o.clamping_thres()[0] and g.getClampingThresMax() == o.clamping_thres()[1]
    for s in range(3):
        origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=scans.lidar_pose(s), seed=3 + s, colored=color)
        _gpu_insert(g, origin, xyz, rgb if color else None, max_range=12.0, discrete=True)
        o.insert(origin, xyz, rgb if color else None, max_range=12.0, discrete=True)
        c = np.array(origin)
        for md in (0, 1, 2, 0):
            ext = np.array([0.45, 0.45, 0.6]) * (1 + md)
            g.setValueVolume(c - ext, c + ext, g.getClampingThresMin(), md)
            o.setValueVolume(c - ext, c + ext, o.clamping_thres()[0], md)
            what = f"scan {s} min_depth {md}"
            assert same_dump(g.leaves(True), o.leaves(True)), what + ": leaves differ"
            assert same_dump(g.inner(), o.inner()), what + ": inner nodes differ"
        assert g.write() == o.write()
    # fresh map: the volume expands unknown space from the root down
    g2, o2 = _maps(color=color, resolution=0.16)
    g2.setValueVolume([-0.5, -0.3, -0.2], [0.7, 0.4, 0.9], 0.3, 0)
    o2.setValueVolume([-0.5, -0.3, -0.2], [0.7, 0.4, 0.9], 0.3, 0)
    assert same_dump(g2.leaves(True), o2.leaves(True)) and same_dump(g2.inner(), o2.inner())
    # no-ops: outside the map, min_depth beyond the tree
    before = g2.write()
    g2.setValueVolume([1e7, 1e7, 1e7], [2e7, 2e7, 2e7], 0.2, 0)
    g2.setValueVolume([-1, -1, -1], [1, 1, 1], 0.2, 40)
    assert g2.write() == before
    # min_depth == depth_levels: the root itself
    g2.setValueVolume([-1, -1, -1], [1, 1, 1], 0.2, 16)
    o2.setValueVolume([-1, -1, -1], [1, 1, 1], 0.2, 16)
    assert same_dump(g2.leaves(True), o2.leaves(True)) and same_dump(g2.inner(), o2.inner())


@pytest.mark.parametrize("half", [1.2, 1.6, 3.0])
def test_set_value_volume_by_size(half):
    """setValueVolume at min_depth 0 takes one of three forms by the number of nodes the volume can meet (ufomap_hip.hip:
    ufomap_map_set_value_volume_ch): the records of all levels in LDS (k_vol_small, <= 1024: a 1.2 m half size is ~980), one workgroup
    level by level (k_vol_all, <= 8192: 1.6 m is ~2250), a launch per level and direction (3 m: ~9700) -- each against the reference's
    recursion, on a map with scans in it (nothing created, values and summaries change) and on a fresh one (everything created)."""
    from ufomap_amd import scans
    g, o = _maps(color=False, resolution=0.16)
    for s in range(2):
        origin, xyz, _ = scans.lidar64(beams=32, azimuths=512, origin=scans.lidar_pose(s), seed=11 + s)
        _gpu_insert(g, origin, xyz, None, max_range=12.0, discrete=True)
        o.insert(origin, xyz, None, max_range=12.0, discrete=True)
        c = np.array(origin) + np.array([0.07, -0.05, 0.03])
        for val in (g.getClampingThresMin(), 0.5):
            g.setValueVolume(c - half, c + half, val, 0)
            o.setValueVolume(c - half, c + half, val, 0)
            assert same_dump(g.leaves(True), o.leaves(True)), f"scan {s} value {val}: leaves differ"
            assert same_dump(g.inner(), o.inner()), f"scan {s} value {val}: inner nodes differ"
    assert g.write() == o.write()
    g2, o2 = _maps(color=False, resolution=0.16)
    g2.setValueVolume([-half, -half + 0.1, -half], [half, half, half - 0.2], 0.3, 0)
    o2.setValueVolume([-half, -half + 0.1, -half], [half, half, half - 0.2], 0.3, 0)
    assert same_dump(g2.leaves(True), o2.leaves(True)) and same_dump(g2.inner(), o2.inner())


@pytest.mark.parametrize("color", [False, True])
def test_point_queries(color):
    """batched getState / contains* / log-odds queries against the oracle (which is pinned on the
    reference's own query functions), after scans and a robot clearing; near surfaces, free, unknown, far away."""
    from ufomap_amd import scans
    g, o = _maps(color=color, resolution=0.16)
    for s in range(3):
        origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=scans.lidar_pose(s), seed=3 + s, colored=color)
        _gpu_insert(g, origin, xyz, rgb if color else None, max_range=12.0, discrete=True)
        o.insert(origin, xyz, rgb if color else None, max_range=12.0, discrete=True)
    c = np.array(origin)
    g.setValueVolume(c - 0.5, c + 0.5, g.getClampingThresMin(), 1)
    o.setValueVolume(c - 0.5, c + 0.5, o.clamping_thres()[0], 1)
    rng = np.random.default_rng(0)
    q = np.concatenate([xyz[::5] + rng.normal(0, 0.05, xyz[::5].shape), rng.uniform(-15, 15, (20000, 3)), rng.uniform(-4000, 4000, (2000, 3))])
    for depth in (0, 1, 2, 5, 14, 15):
        a, b = g.query(q, depth), o.query(q, depth)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"depth {depth}: log-odds differ"
        assert np.array_equal(a[1], b[1]), f"depth {depth}: states differ"
    assert len(set(g.query(q, 0)[1] & 7)) == 3
    assert g.isOccupied(xyz[0]) == bool(o.query(xyz[0])[1][0] & 1)
    # fresh map: everything is the unknown root
    g2, _ = _maps(color=color, resolution=0.16)
    lo, st = g2.query(q[:100], 0)
    assert not lo.any() and set(st) == {4 | 16}


@pytest.mark.parametrize("name", golden_util.server_loop_names())
def test_gpu_matches_server_loop_golden(name):
    """The server's per-message sequence on the GPU -- fused PointCloud2 ingest + discrete integration, robot
    clearing, point queries -- against a recorded run of the unmodified reference (needs no oracle at run time)."""
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    g = golden_util.ServerLoop(name)
    params = dict(g.params)
    color = params.pop("color", False)
    m = (OccupancyMapColor if color else OccupancyMap)(**params)
    for st in g.steps():
        m.insertPointCloud2(st["t"], st["q"], st["data"], st["step"], st["off_xyz"], st["off_rgb"] if color else None,
                            max_range=st["max_range"])
        assert m.getClampingThresMin() == st["clear_value"]
        m.setValueVolume(st["clear_min"], st["clear_max"], st["clear_value"], st["clear_depth"])
    g.check_map(m)
    g.check_queries(m.query)


@pytest.mark.parametrize("async_", [False, True])
def test_predicted_grid_scans_and_their_repeats(async_):
    """Depth-0 scans are enqueued on the ray grid predicted from the previous scan (no read-back of the boxes in
    the middle of the scan); a scan that does not fit flags itself, leaves the map alone and is repeated when it is
    joined. A sensor that stands still, drifts within the margin and jumps by metres: same map as the oracle after
    every scan, with predictions used and repeats happening."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    offsets = [(0, 0, 0), (0, 0, 0), (0.1, 0.05, 0), (0.3, -0.2, 0.05), (3.0, 2.0, 0.3), (3.0, 2.0, 0.3), (-2.5, 1.0, 0.0), (-2.4, 1.1, 0.0)]
    for i, off in enumerate(offsets):
        origin, xyz, _ = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + np.array(off)), seed=50 + i)
        _gpu_insert(g, origin, xyz, max_range=10.0 + (i % 3), discrete=bool(i & 1), async_=async_)
        o.insert(origin, xyz, max_range=10.0 + (i % 3), discrete=bool(i & 1))
        if not async_:
            _assert_same_map(g, o, f"scan {i}")
    g.insertPointCloudWait()
    _assert_same_map(g, o, "final")
    d = g.debug()
    assert d[62] >= 4, "predictions were not used"
    assert 1 <= d[63] < d[62], "the jumps should have forced repeats (and only some scans)"
    # with predictions switched off: the same map
    g2, _ = _maps(resolution=0.16)
    g2.set_option("spec", 0)
    for i, off in enumerate(offsets):
        origin, xyz, _ = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + np.array(off)), seed=50 + i)
        _gpu_insert(g2, origin, xyz, max_range=10.0 + (i % 3), discrete=bool(i & 1))
    assert same_dump(g2.leaves(True), g.leaves(True)) and same_dump(g2.inner(), g.inner())
    assert g2.debug()[62] == 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_operation_sequence(seed):
    """Everything interleaved as a caller might: sync and pipelined inserts (continuous / discrete, insert depth
    0-2, plain and fused-ingest), robot clearing at depth 0-2, batches of update lists from a second map, point
    queries -- with a sensor that drifts and sometimes jumps (predicted grids fit, then do not). The oracle runs
    the reference's sequence of calls; compared at checkpoints and at the end, byte stream included."""
    import torch
    import oracle
    from ufomap_amd import scans, OccupancyMap
    rng = np.random.default_rng(seed)
    g, o = _maps(resolution=0.16)
    scanner = OccupancyMap(0.16)
    pos = np.array(scans.lidar_pose(0), dtype=np.float64)
    keep = []  # device buffers of pipelined inserts stay alive until joined
    for step in range(28):
        pos = pos + (rng.uniform(-3, 3, 3) * [1, 1, 0.1] if rng.random() < 0.15 else rng.uniform(-0.08, 0.08, 3) * [1, 1, 0.2])
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=tuple(pos), seed=1000 * seed + step)
        mr = float(rng.choice([6.0, 9.0, 12.0]))
        op = rng.choice(["insert", "insert", "insert", "async", "async", "pc2", "clear", "batch", "depth"])
        if op in ("insert", "async"):
            disc = bool(rng.integers(0, 2))
            if op == "async":
                d = torch.from_numpy(xyz).cuda()
                keep.append(d)
                g.insert_device(origin, d.data_ptr(), None, xyz.shape[0], mr, 0, discrete=disc, async_=True)
            else:
                _gpu_insert(g, origin, xyz, max_range=mr, discrete=disc)
            o.insert(origin, xyz, max_range=mr, discrete=disc)
        elif op == "depth":
            dep = int(rng.integers(1, 3))
            _gpu_insert(g, origin, xyz, max_range=mr, depth=dep, discrete=True)
            o.insert(origin, xyz, max_range=mr, depth=dep, discrete=True)
        elif op == "pc2":
            f = (xyz - origin).astype(np.float32)
            f[::31, int(rng.integers(0, 3))] = np.nan
            buf = np.zeros((f.shape[0], 16), np.uint8)
            buf[:, 0:12] = f.view(np.uint8).reshape(-1, 12)
            q = np.array([1.0, 0.0, 0.0, 0.0])
            g.insertPointCloud2(origin, q, buf, 16, (0, 4, 8), None, max_range=mr)
            cx, _ = oracle.ingest(buf, 16, (0, 4, 8), None, q, origin, "port")
            o.insert(origin, cx, max_range=mr, discrete=True)
        elif op == "clear":
            md = int(rng.integers(0, 3))
            ext = rng.uniform(0.3, 0.9, 3)
            g.setValueVolume(pos - ext, pos + ext, g.getClampingThresMin(), md)
            o.setValueVolume(pos - ext, pos + ext, o.clamping_thres()[0], md)
        elif op == "batch":
            bufs, infos = [], []
            for k in range(2):
                o2, x2, _ = scans.lidar64(beams=16, azimuths=256, origin=tuple(pos + [0.05 * k, 0, 0]), seed=77 * seed + 10 * step + k)
                d = torch.from_numpy(x2).cuda()
                info = scanner.scan_keys(o2, d.data_ptr(), x2.shape[0], mr, 0, True)
                b = torch.empty((info.n_hit + info.n_miss) * 16, dtype=torch.uint8, device="cuda")
                scanner.get_keys(b.data_ptr(), b.numel() // 16, info)
                bufs.append(b)
                infos.append(info)
                o.insert(o2, x2, max_range=mr, discrete=True)
            g.apply_keys_batch([b.data_ptr() for b in bufs], infos)
        if step % 9 == 8:
            g.insertPointCloudWait()
            keep.clear()
            assert same_dump(g.leaves(True), o.leaves(True)), f"seed {seed} step {step} ({op}): leaves differ"
            assert same_dump(g.inner(), o.inner()), f"seed {seed} step {step} ({op}): inner nodes differ"
            qs = np.concatenate([xyz[::11], rng.uniform(-12, 12, (500, 3)) + pos])
            a, b = g.query(qs, int(rng.integers(0, 4))), None
            d = int(rng.integers(0, 4))
            a, b = g.query(qs, d), o.query(qs, d)
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]), f"seed {seed} step {step}: queries differ"
    g.insertPointCloudWait()
    assert same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner()), f"seed {seed}: final map differs"
    assert g.write() == o.write()


def test_batch_integrator_rccl_world1():
    """BatchIntegrator on HBM tensors through the nccl (RCCL) backend with a single rank: the same code
    path the 8-GPU run takes, minus the peers."""
    import os
    import torch
    import torch.distributed as dist
    from ufomap_amd import OccupancyMap, scans
    from ufomap_amd.dist import BatchIntegrator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g, o = _maps(resolution=0.16)
        bi = BatchIntegrator(g, dist.group.WORLD, torch.device("cuda", 0))
        for s in range(3):
            origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, beams=32, azimuths=1024)
            d = torch.from_numpy(xyz).cuda()
            bi.integrate(origin, d.data_ptr(), xyz.shape[0], 20.0, 0, True)
            o.insert(origin, xyz, max_range=20.0, discrete=True)
        assert same_dump(g.leaves(True), o.leaves(True))
        assert same_dump(g.inner(), o.inner())
    finally:
        dist.destroy_process_group()


RAY_KERNELS = {
    "cast": {},                                        # k_cast: set-up + uniform segments + walk (default)
    "cast_k8_64wgs": {"cast_k": 8, "cast_wgs": 64},    # many short segments, several rounds per workgroup
    "walk_1lane": {"cast": 0, "dda_lanes": 1},         # k_ray_setup + k_walk (bit grid in LDS)
    "walk_4lanes": {"cast": 0, "dda_lanes": 4},
    "seg_bytes_lds": {"dda_bits": 0},                  # k_dda_seg<DDA_LDSGRID> (byte-per-block grid in LDS)
    "seq_bytes_lds": {"dda_bits": 0, "dda_seg": 0},    # k_dda<false, DDA_LDSGRID>, one lane per ray
    "cast_box": {"cast_global": 2, "spec": 0},         # k_cast<2>: bit grid in HBM, a workgroup's box of it in LDS (grids beyond LDS)
    "cast_box_k8_64wgs": {"cast_global": 2, "spec": 0, "cast_k": 8, "cast_wgs": 64},
    "cast_box_fallback": {"cast_global": 4, "spec": 0},  # ... boxes that do not fit: marks one by one through the LDS filter
    "cast_global": {"cast_global": 3, "spec": 0},      # k_cast<1>: every mark through the LDS filter
    "filter": {"dda_mode": 1},                         # k_dda_seg<DDA_FILTER>
    "direct": {"dda_mode": 2},                         # k_dda<false, DDA_DIRECT>
}


@pytest.mark.parametrize("variant", sorted(RAY_KERNELS))
def test_all_ray_kernel_modes_agree(variant):
    """Every variant of the ray kernel (fused / segmented / sequential; LDS bit grid, LDS byte grid, LDS
    filter, direct atomics) marks the same cells in the same number of steps: the oracle's."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    for k, v in RAY_KERNELS[variant].items():
        g.set_option(k, v)
    mode = variant
    origin, xyz, _ = scans.lidar64(beams=32, azimuths=1024)
    for _ in range(2):
        _gpu_insert(g, origin, xyz, max_range=20.0, discrete=False)
        o.insert(origin, xyz, max_range=20.0, discrete=False)
    assert np.array_equal(g.last_misses(), o.last_misses())
    assert g.last_counts()["steps"] == o.last_steps()
    _assert_same_map(g, o, f"dda_mode {mode}")


def test_update_list_count_pass_and_table_growth():
    """A grid larger than the list-size guess takes the counting pass (exact update-list size); the node
    table grows by re-hash several times on the way. Result unchanged."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.08)
    g.set_option("entry_guess", 1000)
    origin, xyz, _ = scans.lidar64(beams=32, azimuths=1024)
    for _ in range(2):
        _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
    assert g.last_counts()["blocks_touched"] > 1000
    assert g.stats()["inner_nodes"] > (1 << 16) // 2  # more live blocks than the initial table could hold at load 0.6
    _assert_same_map(g, o, "retry")


def _c3_depth0_properties(width, height, full_export):
    from ufomap_amd import OccupancyMap, scans
    m = OccupancyMap(0.002)
    origin, xyz, _ = scans.rgbd(width=width, height=height)
    _gpu_insert(m, origin, xyz, max_range=5.0, discrete=True)
    c = m.last_counts()
    # unique hit voxels = distinct floor(x/res) triples of the in-range points, computed independently
    keys = np.floor(xyz * (1.0 / 0.002)).astype(np.int64)
    uniq = np.unique(keys, axis=0).shape[0]
    assert c["hits"] == uniq == c["rays"] and c["oob_dropped"] == 0
    st1 = m.stats()
    if full_export:
        codes, depths, occ, _ = m.leaves()
        hit, miss = np.float32(np.log(0.7 / 0.3)), np.float32(np.log(0.4 / 0.6))
        assert set(np.unique(occ).tolist()) <= {float(miss), float(np.float32(hit + miss))}  # one scan, fresh map
        assert int((occ > 0).sum()) == uniq and np.all(depths[occ > 0] == 0)  # every hit voxel: depth-0 leaf, hit+miss
        n_free_vox = int((np.uint64(8) ** depths[occ < 0].astype(np.uint64)).sum())
        assert n_free_vox >= len(m.last_misses()) - uniq  # free space, partly collapsed into coarser leaves
    # the same scan again changes values only: same node blocks, same leaves
    _gpu_insert(m, origin, xyz, max_range=5.0, discrete=True)
    assert m.last_counts()["blocks_created"] == 0 or m.stats()["inner_nodes"] == st1["inner_nodes"]
    st2 = m.stats()
    assert (st2["inner_nodes"], st2["leaf_nodes"]) == (st1["inner_nodes"], st1["leaf_nodes"])
    return c


def test_c3_depth0_small_frame_properties():
    """2 mm / 5 m RGB-D at insert depth 0 is beyond what the CPU oracle finishes in a test: size-independent
    properties on a 160x120 frame (full leaf export)."""
    c = _c3_depth0_properties(160, 120, True)
    assert c["steps"] > 1e7


def test_full_size_c3_depth0_counts():
    """BASELINE config C3 at insert depth 0, full size: 307 200 points, ~4.7e8 DDA steps, ~5e7 node blocks
    (the reference needs 85-168 s and 20 GB for this scan)."""
    c = _c3_depth0_properties(640, 480, False)
    assert c["steps"] > 3e8


def test_pipelined_async_inserts_equal_sequential():
    """async=True: the scan half of scan i+1 overlaps the map half of scan i on a second stream (the
    reference overlaps its head loop with the previous integration, OMB:315). Same map as sequential."""
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    clouds = []
    for s in range(8):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, beams=32, azimuths=1024)
        clouds.append((origin, xyz))
        o.insert(origin, xyz, max_range=20.0, discrete=True)
    for origin, xyz in clouds:
        _gpu_insert(g, origin, xyz, max_range=20.0, discrete=True, async_=True)  # host buffers: copied before return
    g.insertPointCloudWait()
    assert g.insertPointCloudDone()
    _assert_same_map(g, o, "pipelined")
    assert g.last_counts()["points"] == clouds[-1][1].shape[0]


def test_write_empty_map_and_file(tmp_path):
    """Octree::write on a fresh map (root = one unknown leaf) and through a file."""
    g, o = _maps(resolution=0.16)
    assert g.write() == o.write()
    gc, oc = _maps(color=True, resolution=0.08)
    assert gc.write() == oc.write()
    p = tmp_path / "map.ufo"
    g.write(str(p))
    assert p.read_bytes() == o.write() and p.read_bytes().startswith(b"# UFOMap file")
