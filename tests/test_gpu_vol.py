"""Round 4 (run with -m gpu on an MI355X): the VOLUME path (vol_kernels.h) -- depth-0 scans whose ray grid is beyond the
steady-state path's, e.g. a 2 mm RGB-D frame: ray cells and hit voxels in tile-major brick grids (k_vhits, k_vdda, k_vlist),
then the tiled tree update over the listed tiles (k_tile<VOL>, k_up level after level, k_ftail). Against the reference scan by
scan: values, flags, leaf structure under pruning, byte stream; the marking stage against the port's hit / miss codes."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library)

from conftest import same_dump

pytestmark = pytest.mark.gpu


def _kind():
    import oracle
    return "reference" if oracle.available("reference") else "port"


def _maps(kind="port", **params):
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap
    return OccupancyMap(**params), OracleMap(kind=kind, **params)


def _insert(g, origin, xyz, max_range, discrete, async_=False):
    from ufomap_amd import PointCloud
    (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), max_range, 0, False, 0, async_)


def _assert_same_map(g, o, what=""):
    gl, ol = g.leaves(True), o.leaves(True)
    assert len(gl[0]) == len(ol[0]), f"{what}: leaf count {len(gl[0])} vs oracle {len(ol[0])}"
    assert np.array_equal(gl[0], ol[0]) and np.array_equal(gl[1], ol[1]), f"{what}: leaf codes/depths differ"
    assert np.array_equal(gl[2], ol[2]), f"{what}: log-odds differ"
    assert same_dump(g.inner(), o.inner()), f"{what}: inner-node dump differs"
    assert g.write() == o.write(), f"{what}: map byte stream differs"


def _force_vol(g):
    g.set_option("vol", 2)   # also for ray grids the steady-state path would take
    g.set_option("spec", 0)  # ... which is chosen before the boxes are known: no predicted grids


def _wander(n_scans, beams=32, azimuths=512, spread=1.0, seed0=300):
    from ufomap_amd import scans
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    rng = np.random.default_rng(seed0)
    out = []
    for i in range(n_scans):
        off = rng.uniform(-spread, spread, 3) * [1, 1, 0.1]
        out.append(scans.lidar64(beams=beams, azimuths=azimuths, origin=tuple(base + off), seed=seed0 + i)[:2])
    return out


@pytest.mark.parametrize("discrete,res", [(True, 0.16), (False, 0.16), (True, 0.08)])
def test_volume_path_scan_by_scan(discrete, res):
    """A wandering LiDAR forced onto the volume path: voxels are hit, missed, saturate, collapse and are re-expanded; the map
    equals the reference's after every few scans, the marking stage the port's hit and miss codes."""
    g, o = _maps(kind=_kind(), resolution=res)
    _, p = _maps(kind="port", resolution=res)
    _force_vol(g)
    n_scans = 9 if res >= 0.16 else 6  # (the CPU checkers set this test's duration)
    for i, (origin, xyz) in enumerate(_wander(n_scans)):
        _insert(g, origin, xyz, 12.0, discrete, async_=bool(i & 1))
        o.insert(origin, xyz, max_range=12.0, discrete=discrete)
        p.insert(origin, xyz, max_range=12.0, discrete=discrete)
        g.insertPointCloudWait()
        assert np.array_equal(g.last_hits(), p.last_hits()), f"scan {i}: hit voxels differ"
        assert np.array_equal(g.last_misses(), p.last_misses()), f"scan {i}: ray cells differ"
        assert g.last_counts()["steps"] == p.last_steps(), f"scan {i}: DDA step count differs"
        if i in (0, 1, 5, n_scans - 1):
            _assert_same_map(g, o, f"after scan {i}")
    d = g.debug()
    assert d[50] == n_scans and d[61] == 0, f"the scans did not take the volume path: {d[48:51]}, fast {d[61]}"


def test_volume_path_saturation_and_pruning():
    """The same two scans over and over: everything saturates, subtrees collapse and are re-expanded (the last-update chain through
    k_tile, k_up on two levels and k_ftail)."""
    from ufomap_amd import scans
    g, o = _maps(kind=_kind(), resolution=0.16)
    _force_vol(g)
    poses = [scans.lidar_pose(0), tuple(np.array(scans.lidar_pose(0)) + [0.35, -0.2, 0.0])]
    clouds = [scans.lidar64(beams=32, azimuths=512, origin=p, seed=7 + k)[:2] for k, p in enumerate(poses)]
    for i in range(16):
        origin, xyz = clouds[i & 1]
        _insert(g, origin, xyz, 8.0, True)
        o.insert(origin, xyz, max_range=8.0, discrete=True)
        if i in (1, 9, 15):
            _assert_same_map(g, o, f"after scan {i}")
    assert g.debug()[50] == 16


def test_volume_path_grows_the_table_in_the_middle_of_a_walk():
    """With the up-front growth switched off the first walk runs out of its reserve: tiles stand back, the table is exchanged, the
    tiles that are left are run -- more than once as the map grows; the map stays the reference's."""
    from ufomap_amd import scans
    g, o = _maps(kind=_kind(), resolution=0.08)
    _force_vol(g)
    g.set_option("vol_pregrow", 0)
    for s in range(3):
        origin, xyz, _ = scans.lidar64(beams=32, azimuths=1024, origin=scans.lidar_pose(s), seed=100 + s)
        _insert(g, origin, xyz, 20.0, True)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
        _assert_same_map(g, o, f"scan {s}")
    d = g.debug()
    assert d[50] == 3 and d[49] >= 1, f"no growth in the middle of a walk: {d[48:51]}"


def test_volume_path_takes_the_rgbd_frame_by_itself():
    """BASELINE configs[2] at insert depth 0 on a reduced frame (160 x 120 pixels, 2 mm: rays of up to 1 500 cells, 1.1e8 leaves):
    nothing forced -- the scan's box is beyond the steady-state path. Fresh and warm against the fingerprints of the unmodified
    reference's dumps (tests/golden/digests.json: c3_depth0_160x120; leaf for leaf against the checker itself:
    test_gpu_parity2.py::test_c3_depth0_reduced_frame_leaf_for_leaf, which takes this path as well), and what only this path
    promises: no growth in the middle of a walk, no device allocation in a warm scan."""
    from ufomap_amd import OccupancyMap, capi, scans
    import golden_util
    fx = golden_util.digests()["c3_depth0_160x120"]
    assert fx["scans"][0][1] == dict(width=160, height=120) and fx["params"] == dict(resolution=0.002)
    g = OccupancyMap(resolution=0.002)
    origin, xyz, _ = scans.rgbd(width=160, height=120)
    for i in range(2):
        _insert(g, origin, xyz, 5.0, True)
        assert g.digest() == tuple(int(v) for v in fx["steps"][i]["digest"]), f"scan {i}: digest differs from the reference's"
    d = g.debug()
    assert d[50] == 2 and d[49] == 0, f"volume path scans {d[50]}, growths in a walk {d[49]}"
    a0 = capi.alloc_counters()
    _insert(g, origin, xyz, 5.0, True)
    a1 = capi.alloc_counters()
    assert a1["mallocs"] == a0["mallocs"] and a1["rehashes"] == a0["rehashes"], "a warm scan allocated device memory"


def test_general_path_and_volume_path_agree_on_a_clipped_scan():
    """A scan with a ray that is clipped at the map cube (a return far outside a small map) turns to the general path by itself:
    the same map as a handle that never tries the volume path (and the checker's: the segment enters through a min face)."""
    from ufomap_amd import OccupancyMap, scans
    g, o = _maps(kind=_kind(), resolution=0.04, depth_levels=11)  # a 41 m cube
    g2 = OccupancyMap(resolution=0.04, depth_levels=11)
    g2.set_option("vol", 0)
    g2.set_option("spec", 0)
    _force_vol(g)
    origin, xyz, _ = scans.lidar64(beams=16, azimuths=256)
    xyz = xyz.copy()
    xyz[5] = [-60.0, 3.0, 1.0]
    _insert(g, origin, xyz, -1.0, False)
    _insert(g2, origin, xyz, -1.0, False)
    o.insert(origin, xyz, max_range=-1.0, discrete=False)
    assert g.debug()[48] + g.debug()[50] == 1  # (a segment clipped onto a min face is an ordinary ray; one clipped so that a key leaves the range turns back)
    assert same_dump(g.leaves(True), g2.leaves(True)) and same_dump(g.inner(), g2.inner())
    _assert_same_map(g, o, "clipped")


@pytest.mark.parametrize("discrete", [True, False])
def test_simple_ray_casting_on_the_steady_state_path(discrete):
    """`simple_ray_casting` (freeSpaceSimple, occupancy_map_base.h:1303-1339; a switch of the reference's server) on the
    steady-state path (k_fcast_simple): a wandering sensor, synchronous and pipelined calls, normal and fixed-step scans mixed in
    one map -- against the reference scan by scan; the fast-path counter shows where the scans went."""
    from ufomap_amd import PointCloud
    g, o = _maps(kind=_kind(), resolution=0.16)
    _, p = _maps(kind="port", resolution=0.16)
    n_simple = 0
    for i, (origin, xyz) in enumerate(_wander(16, spread=0.4)):
        simple = i % 5 != 4
        n_simple += simple
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), 12.0, 0, simple, 0, bool(i & 1))
        o.insert(origin, xyz, max_range=12.0, discrete=discrete, simple_ray_casting=simple)
        p.insert(origin, xyz, max_range=12.0, discrete=discrete, simple_ray_casting=simple)
        if i in (1, 2, 7, 15):
            g.insertPointCloudWait()
            _assert_same_map(g, o, f"after scan {i}")
            assert np.array_equal(g.last_misses(), p.last_misses()), f"scan {i}: ray cells differ"
            assert g.last_counts()["steps"] == p.last_steps()
    d = g.debug()
    assert d[61] >= 12, f"the scans did not take the steady-state path: {d[61]} of 16"
    g2, _ = _maps(kind="port", resolution=0.16)
    g2.set_option("fast_simple", 0)
    for i, (origin, xyz) in enumerate(_wander(16, spread=0.4)):
        (g2.insertPointCloudDiscrete if discrete else g2.insertPointCloud)(origin, PointCloud(xyz), 12.0, 0, i % 5 != 4, 0, False)
    assert same_dump(g.leaves(True), g2.leaves(True)) and same_dump(g.inner(), g2.inner()), "general path and steady-state path differ"


@pytest.mark.parametrize("early", [1, 3, 10])
@pytest.mark.parametrize("mode", ["continuous", "discrete", "simple", "discrete_d1", "discrete_sparse", "continuous_sparse_tiny"])
def test_early_stopping(early, mode):
    """`early_stopping` > 0 (occupancy_map_base.h:1289-1298, 1327-1333): a ray ends once that many cells in a row were in the
    scan's set already -- put there by rays cast EARLIER. The device finds every ray's stop as the fixed point of "who visits a
    cell first" (scan_kernels.h: k_es_mark / k_es_stops): same ray cells, step count and map as the reference casting the rays one
    after the other; sweeps (neighbouring rays share most cells), sync and async calls, mixed with ordinary scans."""
    from ufomap_amd import PointCloud, scans
    # (*_sparse: "who visits a cell first" in the hash of visited cells that ray boxes beyond the scratch limit take -- round 6; _tiny: a
    # set that starts far too small and has to be doubled several times inside the first round)
    sparse = mode.endswith("_sparse") or mode.endswith("_sparse_tiny")
    kw = dict(continuous=dict(), discrete=dict(discrete=True), simple=dict(discrete=True, simple_ray_casting=True), discrete_d1=dict(discrete=True, depth=1),
              discrete_sparse=dict(discrete=True), continuous_sparse_tiny=dict())[mode]
    g, o = _maps(kind=_kind(), resolution=0.16)
    if sparse:
        g.set_option("es_sparse", 2 if mode.endswith("_tiny") else 1)
    _, p = _maps(kind="port", resolution=0.16)
    rounds = []
    for s in range(5):
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose(s % 3), seed=100 + s)
        es = 0 if s == 3 else early
        ins = g.insertPointCloudDiscrete if kw.get("discrete") else g.insertPointCloud
        ins(origin, PointCloud(xyz), 12.0, kw.get("depth", 0), kw.get("simple_ray_casting", False), es, bool(s & 1))
        o.insert(origin, xyz, max_range=12.0, early_stopping=es, **kw)
        p.insert(origin, xyz, max_range=12.0, early_stopping=es, **kw)
        g.insertPointCloudWait()
        assert np.array_equal(g.last_misses(), p.last_misses()), f"scan {s}: ray cells differ"
        assert g.last_counts()["steps"] == p.last_steps(), f"scan {s}: cells visited differ"
        if es:
            rounds.append(g.debug()[47])
    _assert_same_map(g, o, mode)
    assert all(1 <= r <= 64 for r in rounds), f"rounds to settle: {rounds}"


@pytest.mark.parametrize("early", [1, 3])
def test_early_stopping_on_a_ray_box_beyond_the_dense_array(early):
    """VERDICT r5 (missing 3): a 2 mm RGB-D frame at insert depth 0 spans 1 500 x 1 800 x 1 400 cells -- 3.8e9, beyond a dense
    first-ray array (2^32 entries, 15 GB) -- and the reference has no such limit (occupancy_map_base.h:1289-1298). The stops now settle
    on a sparse set of the cells the rays visit (scan_kernels.h: EsArgs::hkeys), grown inside the first round: a reduced frame (80 x 60
    pixels: the same box, 4 800 rays of ~1 500 cells) against the reference -- map, ray cells' count and steps."""
    from ufomap_amd import OccupancyMap, PointCloud, scans
    import golden_util
    origin, xyz, _ = scans.rgbd(width=80, height=60)
    g, o = _maps(kind=_kind(), resolution=0.002)
    _, p = _maps(kind="port", resolution=0.002)
    g.insertPointCloudDiscrete(origin, PointCloud(xyz), 5.0, 0, False, early, False)
    o.insert(origin, xyz, max_range=5.0, discrete=True, early_stopping=early)
    p.insert(origin, xyz, max_range=5.0, discrete=True, early_stopping=early)
    assert g.last_counts()["steps"] == p.last_steps(), "cells visited differ"
    assert g.digest() == tuple(golden_util.dump_digest(o.leaves(True), o.inner())), "the map differs from the reference's"
    assert 1 <= g.debug()[47] <= 64, f"rounds to settle: {g.debug()[47]}"


def _frames_for_cuts():
    """Scans whose rays differ in what the cuts have to get right: random directions and lengths (every dominant axis, either sign,
    ties between axes on the diagonals), a LiDAR's bundles, an RGB-D frame's long parallel rays."""
    from ufomap_amd import scans
    o1, x1, _ = scans.random_cloud(20000, seed=11, extent=9.0)
    x1 = x1.copy()
    diag = np.array([[sx * d, sy * d, sz * d] for d in (1.28, 3.2, 5.12, 7.04) for sx in (-1, 1) for sy in (-1, 0, 1) for sz in (-1, 1)])
    x1[:len(diag)] = o1 + diag  # (diagonals: t_max ties between the axes all the way)
    o2, x2, _ = scans.lidar64(beams=32, azimuths=1024, origin=scans.lidar_pose(2), seed=5)
    o3, x3, _ = scans.rgbd(width=160, height=120)
    return [("random", 0.08, o1, x1, -1.0), ("lidar", 0.08, o2, x2, 15.0), ("rgbd", 0.004, o3, x3, 5.0)]


@pytest.mark.parametrize("discrete", [True, False])
def test_segmented_ray_walk_marks_the_cells_of_the_sequential_walk(discrete):
    """Round 5: the volume path's rays are cut into segments of ~K cells that lanes walk one each (k_vcutA / k_vcutB / k_vwalk) --
    the cut states rebuilt from the three addition chains. Same ray cells, same step count, same map as the one-lane-per-ray kernel
    (k_vdda, option vol_mode bit 4) for K = 5 ... 192, and as the port's sequential walk (occupancy_map_base.h:1261-1301)."""
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap
    for name, res, origin, xyz, max_range in _frames_for_cuts():
        ref = OccupancyMap(resolution=res)
        _force_vol(ref)
        ref.set_option("vol_mode", 16)
        _insert(ref, origin, xyz, max_range, discrete)
        assert ref.debug()[50] == 1, f"{name}: the scan did not take the volume path"
        cells, steps, dig = ref.last_misses(), ref.last_counts()["steps"], ref.digest()
        if name != "rgbd":
            p = OracleMap(kind="port", resolution=res)
            p.insert(origin, xyz, max_range=max_range, discrete=discrete)
            assert np.array_equal(cells, p.last_misses()) and steps == p.last_steps(), f"{name}: the sequential kernel differs from the port"
        for K in (5, 16, 61, 192):
            g = OccupancyMap(resolution=res)
            _force_vol(g)
            g.set_option("vol_seg", K)
            for rep in range(2):  # (the second scan: the grids and lists of the first have been reused)
                _insert(g, origin, xyz, max_range, discrete)
                assert g.debug()[50] == rep + 1
                assert g.last_counts()["steps"] == steps, f"{name} K={K} scan {rep}: step count {g.last_counts()['steps']} vs {steps}"
                assert np.array_equal(g.last_misses(), cells), f"{name} K={K} scan {rep}: ray cells differ"
                if 0 == rep:
                    assert g.digest() == dig, f"{name} K={K}: maps differ"


@pytest.mark.parametrize("pregrow", [1, 0])
def test_volume_path_pipelined_scans(pregrow):
    """Round 5: an asynchronous call returns with the volume path's walk ENQUEUED; the next call casts its rays (brick grids of its
    own hand-over set) while that walk runs and joins it before its own. Back-to-back asynchronous scans, with and without the
    up-front table growth (without: walks run out of their reserve and are finished by the call that joins them) -- the map is
    the reference's after the same scans one by one."""
    g, o = _maps(kind=_kind(), resolution=0.16)
    _force_vol(g)
    g.set_option("vol_pregrow", pregrow)
    for i, (origin, xyz) in enumerate(_wander(8, spread=2.0)):
        _insert(g, origin, xyz, 12.0, True, async_=True)
        o.insert(origin, xyz, max_range=12.0, discrete=True)
        if i in (4, 7):
            g.insertPointCloudWait()
            _assert_same_map(g, o, f"after scan {i}")
    d = g.debug()
    assert d[50] == 8, f"the scans did not take the volume path: {d[48:51]}"
    if not pregrow:
        assert d[49] >= 1, "no walk ran out of its reserve"


@pytest.mark.parametrize("variant", ["async", "vol_sync", "all_sync"])
def test_volume_path_walk_in_flight_and_other_paths(variant):
    """A volume-path walk still enqueued when scans of the other paths arrive (steady-state path, general path): each is applied
    after it, in order."""
    g, o = _maps(kind=_kind(), resolution=0.16)
    if variant == "vol_sync":
        g.set_option("vol_async", 0)
    seq = _wander(9, spread=0.5)
    for i, (origin, xyz) in enumerate(seq):
        forced = i in (0, 1, 4, 5, 8)
        g.set_option("vol", 2 if forced else 1)
        g.set_option("spec", 0 if forced or i == 6 else 1)  # (scan 6: the general path)
        _insert(g, origin, xyz, 12.0, True, async_=variant != "all_sync")
        o.insert(origin, xyz, max_range=12.0, discrete=True)
        if variant == "all_sync":
            _assert_same_map(g, o, f"scan {i}")
    g.insertPointCloudWait()
    _assert_same_map(g, o, "mixed paths")
    assert g.debug()[50] >= 5


@pytest.mark.parametrize("mode", ["sync", "async", "mixed_clouds"])
def test_colour_maps_on_the_volume_path(mode):
    """Round 5: OccupancyMapColor on the volume path (k_tile<true, VOL>, k_up<true>, k_ftail<true>): a voxel that receives a hit takes
    the colour of its FIRST point (found through the scan's hit hash) blended with what it has (occupancy_map_color.h:195-287,
    occupancy_map_color.cpp:142-171), every node above carries the root mean square of its children's colours, pruning needs equal
    colours -- against the reference scan by scan: leaves with colours, inner nodes, byte stream; a wandering sensor, scans of one
    colour (subtrees collapse), plain clouds into the colour map in between, asynchronous calls."""
    from ufomap_amd import scans, OccupancyMapColor, PointCloud, PointCloudColor
    from oracle import OracleMap
    from test_gpu_batch import _assert_same_colour_map
    g, o = OccupancyMapColor(resolution=0.16), OracleMap(kind=_kind(), color=True, resolution=0.16)
    _force_vol(g)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    rng = np.random.default_rng(78)
    n_scans = 12
    for i in range(n_scans):
        off = rng.uniform(-0.6, 0.6, 3) * [1, 1, 0.1]
        origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + off), seed=600 + i, colored=True)
        if i % 5 == 4:
            rgb = np.full_like(rgb, 90 + i)  # (a scan of one colour: whole subtrees become equal and collapse)
        plain = mode == "mixed_clouds" and i % 3 == 1
        cloud = PointCloud(xyz) if plain else PointCloudColor(xyz, rgb)
        async_ = mode == "async" or (mode == "mixed_clouds" and i % 4 == 3)
        g.insertPointCloudDiscrete(origin, cloud, 12.0, 0, False, 0, async_)
        o.insert(origin, xyz, None if plain else rgb, max_range=12.0, discrete=True)
        if i in (1, 6):
            g.insertPointCloudWait()
            _assert_same_colour_map(g, o, f"{mode}: after scan {i}")
    g.insertPointCloudWait()
    _assert_same_colour_map(g, o, f"{mode}: final")
    d = g.debug()
    assert d[50] == n_scans and d[61] == 0, f"the colour scans did not take the volume path: {d[48:51]}, fast {d[61]}"


def test_coloured_rgbd_frame_on_the_volume_path():
    """The reference's only published figure is a COLOURED map at 2 mm (README.md:10-11). A reduced coloured frame (160 x 120 pixels,
    2 mm, insert depth 0) takes the volume path by itself: fresh and warm against the fingerprints of the unmodified reference's
    dumps (tests/golden/digests.json: c3_colour_160x120)."""
    from ufomap_amd import OccupancyMapColor, PointCloudColor, scans
    import golden_util
    fx = golden_util.digests().get("c3_colour_160x120")
    if fx is None:
        pytest.skip("fixture c3_colour_160x120 not generated")
    g = OccupancyMapColor(resolution=0.002)
    origin, xyz, rgb = scans.rgbd(width=160, height=120, colored=True)
    for i in range(2):
        g.insertPointCloudDiscrete(origin, PointCloudColor(xyz, rgb), 5.0)
        assert g.digest() == tuple(int(v) for v in fx["steps"][i]["digest"]), f"scan {i}: digest differs from the reference's"
    assert g.debug()[50] == 2, "the coloured frame did not take the volume path"


def test_per_xcd_marks_equal_the_general_paths_on_random_frames():
    """The volume path marks one copy of the brick grid PER XCD with atomics that need no more than the XCD's L2 (vol_kernels.h: a
    start-up self-test checks that assumption on the device). 100 random frames with rays in every direction -- the tiles round the
    sensor are marked by waves of all eight XCDs -- cell for cell against the general path's ray cells (k_cast / k_dda: one grid,
    device-scope or LDS marks), both ray kernels of the volume path; the maps stay equal throughout."""
    from ufomap_amd import OccupancyMap, scans
    gv, gs, gg = OccupancyMap(resolution=0.1), OccupancyMap(resolution=0.1), OccupancyMap(resolution=0.1)
    _force_vol(gv)
    _force_vol(gs)
    gs.set_option("vol_mode", 16)  # one lane per ray (k_vdda)
    gg.set_option("vol", 0)
    gg.set_option("spec", 0)
    for seed in range(100):
        origin, xyz, _ = scans.random_cloud(2000 + 37 * seed, seed=900 + seed, extent=3.0 + 0.05 * seed, origin=(0.3 + 0.01 * seed, -0.2, 0.4))
        for g in (gv, gs, gg):
            _insert(g, origin, xyz, -1.0, bool(seed & 1))
        ref = gg.last_misses()
        assert np.array_equal(gv.last_misses(), ref), f"frame {seed}: the segmented walk's ray cells differ from the general path's"
        assert np.array_equal(gs.last_misses(), ref), f"frame {seed}: k_vdda's ray cells differ from the general path's"
    assert gv.debug()[50] == 100 and gs.debug()[50] == 100 and gg.debug()[50] == 0
    assert gv.digest() == gg.digest() == gs.digest()
