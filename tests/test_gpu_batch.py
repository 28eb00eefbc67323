"""Round 3 (run with -m gpu on an MI355X): ONE walk of the tree for a batch of scans (fast_kernels.h: k_tile / k_ftail
over B scans) must leave exactly the map the reference leaves after integrating the scans one after the other
(occupancy_map_base.h:340-417 called B times) -- values, flags, leaf structure under pruning, byte stream."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library: torch brings its own copy of the HIP runtime, and the first one loaded owns the GPU)

from conftest import same_dump

pytestmark = pytest.mark.gpu


def _kind():
    import oracle
    return "reference" if oracle.available("reference") else "port"


def _maps(kind="port", **params):
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap
    return OccupancyMap(**params), OracleMap(kind=kind, **params)


def _insert(g, origin, xyz, max_range, discrete, async_):
    from ufomap_amd import PointCloud
    (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), max_range, 0, False, 0, async_)


def _assert_same_map(g, o, what=""):
    gl, ol = g.leaves(True), o.leaves(True)
    assert len(gl[0]) == len(ol[0]), f"{what}: leaf count {len(gl[0])} vs oracle {len(ol[0])}"
    assert np.array_equal(gl[0], ol[0]) and np.array_equal(gl[1], ol[1]), f"{what}: leaf codes/depths differ"
    assert np.array_equal(gl[2], ol[2]), f"{what}: log-odds differ"
    assert same_dump(g.inner(), o.inner()), f"{what}: inner-node dump differs"
    assert g.write() == o.write(), f"{what}: map byte stream differs"


def _sequence(n_scans, beams=32, azimuths=512, spread=1.0, seed0=300):
    """A sensor that wanders about: the same voxels are hit, missed, saturate, collapse and are re-expanded again and again."""
    from ufomap_amd import scans
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    rng = np.random.default_rng(seed0)
    out = []
    for i in range(n_scans):
        off = rng.uniform(-spread, spread, 3) * [1, 1, 0.1]
        out.append(scans.lidar64(beams=beams, azimuths=azimuths, origin=tuple(base + off), seed=seed0 + i)[:2])
    return out


@pytest.mark.parametrize("batch_max,discrete", [(2, True), (4, True), (16, True), (3, False), (8, False)])
def test_batched_walks_equal_sequential(batch_max, discrete):
    """Pipelined scans whose slots on the map stream are enqueued `hold` at a time, each claim waiting for the newest scan half
    (a test aid: walks over several scans then happen whatever the timing) against the reference integrating them one by
    one -- compared after every few batches and at the end."""
    seq = _sequence(26)
    g, o = _maps(kind=_kind() if batch_max != 4 else "port", resolution=0.16)
    g.set_option("batch_max", batch_max)
    g.set_option("hold", min(batch_max, 7))
    for i, (origin, xyz) in enumerate(seq):
        _insert(g, origin, xyz, 12.0, discrete, True)
        o.insert(origin, xyz, max_range=12.0, discrete=discrete)
        if i in (9, 17):
            g.insertPointCloudWait()
            _assert_same_map(g, o, f"after scan {i}")
            if o.kind == "port":  # (stage outputs -- the last scan's hit / miss codes -- are provided by the port only)
                assert np.array_equal(g.last_hits(), o.last_hits()) and np.array_equal(g.last_misses(), o.last_misses())
    g.insertPointCloudWait()
    _assert_same_map(g, o, "final")
    d = g.debug()
    assert d[61] >= 20, "the scans did not take the fast path"
    assert d[59] == d[61] - d[63], f"every fast-path scan that was not repeated belongs to exactly one walk: {d[59:64]}"
    assert d[60] < d[59], f"no walk took more than one scan ({d[60]} walks, {d[59]} scans)"


def test_batched_walk_saturation_and_pruning():
    """The same scan over and over from two alternating poses: voxels saturate, whole subtrees collapse, are re-expanded by the
    next scan and collapse again -- inside ONE walk (the collapse of scan b is undone by scan b + 1 on the register copy)."""
    from ufomap_amd import scans
    g, o = _maps(kind=_kind(), resolution=0.16)
    g.set_option("batch_max", 8)
    g.set_option("hold", 6)
    poses = [scans.lidar_pose(0), tuple(np.array(scans.lidar_pose(0)) + [0.35, -0.2, 0.0])]
    clouds = [scans.lidar64(beams=32, azimuths=512, origin=p, seed=7 + k)[:2] for k, p in enumerate(poses)]
    for i in range(34):
        origin, xyz = clouds[i & 1]
        _insert(g, origin, xyz, 8.0, True, True)
        o.insert(origin, xyz, max_range=8.0, discrete=True)
        if i in (1, 18):
            g.insertPointCloudWait()
            _assert_same_map(g, o, f"after scan {i}")
    g.insertPointCloudWait()
    _assert_same_map(g, o, "final")
    assert g.debug()[60] < g.debug()[59]


def test_batch_with_a_scan_that_does_not_fit_its_predicted_grid():
    """One scan of a batch jumps by metres: it flags itself, the whole walk stands back, every scan of it is repeated in order."""
    from ufomap_amd import scans
    g, o = _maps(kind=_kind(), resolution=0.16)
    g.set_option("batch_max", 4)
    g.set_option("hold", 4)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    offs = [(0, 0, 0)] * 4 + [(0.1, 0, 0), (0.2, 0.1, 0), (6.0, 4.0, 0.2), (0.2, 0.0, 0), (0.1, 0.1, 0), (6.0, 4.1, 0.2), (6.1, 4.0, 0.2), (0, 0, 0)] + [(0.05, 0, 0)] * 5
    for i, off in enumerate(offs):
        origin, xyz, _ = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + np.array(off)), seed=900 + i)
        _insert(g, origin, xyz, 10.0, True, True)
        o.insert(origin, xyz, max_range=10.0, discrete=True)
    g.insertPointCloudWait()
    _assert_same_map(g, o, "final")
    d = g.debug()
    assert d[63] >= 1, "the jump should have forced a repeat"


def test_fast_path_counter_and_general_path_agree_on_the_bench_sequence():
    """The bench's 25-scan moving-sensor sequence with the steady-state path on and off: same map, and the counters prove the
    five-launch path really ran (a regression of fastEligible or a gate time-out would otherwise pass every other test on the
    general path and only show up as a slow bench)."""
    import torch
    from ufomap_amd import scans, OccupancyMap
    clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
    d_clouds = [torch.from_numpy(c[1]).cuda() for c in clouds]
    digests, counters = [], []
    for fast in (1, 0):
        g = OccupancyMap(0.16)
        g.set_option("fast", fast)
        for i in range(25):
            p = i % 8
            g.insert_device(clouds[p][0], d_clouds[p].data_ptr(), None, d_clouds[p].shape[0], max_range=20.0, depth=0, discrete=True, async_=True)
        g.insertPointCloudWait()
        digests.append(g.digest())
        counters.append(g.debug())
    assert digests[0] == digests[1]
    on, off = counters
    assert on[61] >= 20 and on[59] == on[61] - on[63] and on[60] >= 1, f"fast path did not run: {on[59:64]}"
    assert on[58] == 0, "a stream hand-over timed out"
    assert off[61] == 0 and off[60] == 0


@pytest.mark.parametrize("opts", [{"solo": 0}, {"solo": 1}, {"lazy_done": 0}, {"lazy_done": 1, "cast_wgs": 64}, {"cast_wgs": 256, "cast_batch": 96, "cast_qcap": 256},
                                  {"cast_threads": 1024, "cast_wgs": 100, "cast_batch": 320}, {"cast_prio": 3, "tstamps": 1},
                                  {"cast_fused": 2}, {"cast_fused": 2, "cast_wgs": 7, "cast_batch": 64, "cast_qcap": 128}, {"cast_fused": 1}, {"cast_oct": 0}, {"cast_oct": 0, "cast_fused": 1},
                                  {"cast_oct": 1, "cast_wgs": 9}, {"cast_oct": 1, "cast_threads": 256}, {"cast_oct": 1, "cast_wgs": 40, "cast2_k": 16},
                                  {"cast_oct": 1, "cast_wgs": 700, "cast_threads": 256, "cast2_k": 24}])
def test_scheduling_options_never_change_the_map(opts):
    """How the five kernels are enqueued -- a synchronous call alone on the map stream (solo), the end of a scan half
    published by the next scan's gate kernel (lazy_done), the ray kernel's workgroups / rays per round / segment queue --
    is scheduling only: synchronous, asynchronous and mixed sequences of calls, continuous and discrete, leave the
    reference's map, and the counters say the fast path ran."""
    g, o = _maps(_kind(), resolution=0.16)
    for k, v in opts.items():
        g.set_option(k, v)
    seq = _sequence(14, beams=32, azimuths=512, spread=0.5, seed0=900)
    pattern = [False, False, True, True, True, False, True, True, False, False, True, True, True, True]  # async?
    for i, ((origin, xyz), async_) in enumerate(zip(seq, pattern)):
        discrete = i % 3 != 2
        _insert(g, origin, xyz, 15.0, discrete, async_)
        o.insert(origin, xyz, max_range=15.0, discrete=discrete)
        if not async_ and i in (1, 5, 9):
            assert g.digest() == g.digest(), "digest is stable"
    g.insertPointCloudWait()
    _assert_same_map(g, o, f"options {opts}")
    d = g.debug()
    assert d[61] >= 8 and d[58] == 0, f"fast path did not run / a hand-over timed out: {d[58:64]}"
    if opts.get("tstamps"):
        ts, newest = g.timeline()
        assert newest >= 1 and ts[newest % 4096, 6] > ts[newest % 4096, 2] > 0, "pipeline time stamps of the newest scan"


def _assert_same_colour_map(g, o, what=""):
    gl, ol = g.leaves(True), o.leaves(True)
    assert len(gl[0]) == len(ol[0]), f"{what}: leaf count {len(gl[0])} vs oracle {len(ol[0])}"
    for k in range(len(gl)):
        assert np.array_equal(gl[k], ol[k]), f"{what}: leaf output {k} differs"
    assert same_dump(g.inner(), o.inner()), f"{what}: inner-node dump differs"
    assert g.write() == o.write(), f"{what}: map byte stream differs"


@pytest.mark.parametrize("mode", ["sync", "async", "batched", "mixed_clouds"])
def test_colour_maps_on_the_fast_path(mode):
    """OccupancyMapColor through the tiled tree update (k_tile<true> / k_ftail<true>): the voxel takes the colour of its
    first point blended with what it has (OMC.h:195-287, OMC.cpp:142-171), every node above carries the root mean square
    of its children's colours, pruning needs equal colours (OMC.cpp:115-140) -- against the reference scan by scan:
    leaves with colours, inner nodes, byte stream; synchronous, pipelined, several scans per walk, and colour maps fed
    with plain clouds in between (continuous mode included). The counters prove the fast path ran."""
    from ufomap_amd import scans, OccupancyMapColor, PointCloud, PointCloudColor
    from oracle import OracleMap
    kind = _kind()
    g, o = OccupancyMapColor(resolution=0.16), OracleMap(kind=kind, color=True, resolution=0.16)
    if mode == "batched":
        g.set_option("hold", 4)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    rng = np.random.default_rng(77)
    n_scans = 18
    for i in range(n_scans):
        off = rng.uniform(-0.6, 0.6, 3) * [1, 1, 0.1]
        origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + off), seed=500 + i, colored=True)
        if i % 5 == 4:
            rgb = np.full_like(rgb, 90 + i)  # (a scan of one colour: whole subtrees become equal and collapse)
        plain = mode == "mixed_clouds" and i % 3 == 1
        discrete = True if not plain else (i % 2 == 0)
        cloud = PointCloud(xyz) if plain else PointCloudColor(xyz, rgb)
        async_ = mode in ("async", "batched") or (mode == "mixed_clouds" and i % 4 == 3)
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, cloud, 12.0, 0, False, 0, async_)
        o.insert(origin, xyz, None if plain else rgb, max_range=12.0, discrete=discrete)
        if i in (2, 9):
            g.insertPointCloudWait()
            _assert_same_colour_map(g, o, f"{mode}: after scan {i}")
    g.insertPointCloudWait()
    _assert_same_colour_map(g, o, f"{mode}: final")
    d = g.debug()
    assert d[61] >= n_scans - 4 and d[58] == 0, f"colour scans did not take the fast path: {d[58:64]}"
    if mode == "batched":
        assert d[60] < d[59], f"no walk took more than one scan ({d[60]} walks, {d[59]} scans)"
    # the same sequence with the fast path off for colour maps: the general path gives the same map
    g2 = OccupancyMapColor(resolution=0.16)
    g2.set_option("fast_color", 0)
    rng = np.random.default_rng(77)
    for i in range(6):
        off = rng.uniform(-0.6, 0.6, 3) * [1, 1, 0.1]
        origin, xyz, rgb = scans.lidar64(beams=32, azimuths=512, origin=tuple(base + off), seed=500 + i, colored=True)
        if i % 5 == 4:
            rgb = np.full_like(rgb, 90 + i)
        g2.insertPointCloudDiscrete(origin, PointCloudColor(xyz, rgb), 12.0)
    assert g2.debug()[61] == 0


@pytest.mark.parametrize("color,mode", [(False, "sync"), (False, "async"), (False, "batched"), (True, "sync"), (True, "batched")])
def test_ray_grids_beyond_lds_on_the_fast_path(color, mode):
    """A scan whose ray grid does not fit the ray kernel's LDS (6 cm voxels, 12 m range: ~1 MB of bits, ~15 000 tiles) still
    takes the tiled tree update: rays through k_fselect / k_cast<2> into a grid in HBM, level 4 of the tree in parallel
    (k_up), k_ftail from level 5 -- against the reference scan by scan, plain and colour maps, continuous scans in between,
    one walk for several scans; and with option big = 0 (the general path) the same map."""
    from ufomap_amd import scans, OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor
    from oracle import OracleMap
    cls = OccupancyMapColor if color else OccupancyMap
    g, o = cls(resolution=0.06), OracleMap(kind=_kind(), color=color, resolution=0.06)
    g2 = cls(resolution=0.06)
    g2.set_option("big", 0)
    if mode == "batched":
        g.set_option("hold", 3)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    rng = np.random.default_rng(5)
    n_scans = 8
    for i in range(n_scans):
        off = rng.uniform(-0.4, 0.4, 3) * [1, 1, 0.1]
        origin, xyz, rgb = scans.lidar64(beams=24, azimuths=384, origin=tuple(base + off), seed=700 + i, colored=True)
        discrete = color or i % 3 != 2
        cloud = PointCloudColor(xyz, rgb) if color else PointCloud(xyz)
        async_ = mode != "sync"
        for m in (g, g2):
            (m.insertPointCloudDiscrete if discrete else m.insertPointCloud)(origin, cloud, 12.0, 0, False, 0, async_)
        o.insert(origin, xyz, rgb if color else None, max_range=12.0, discrete=discrete)
        if i == 4:
            g.insertPointCloudWait()
            (_assert_same_colour_map if color else _assert_same_map)(g, o, f"after scan {i}")
    g.insertPointCloudWait()
    g2.insertPointCloudWait()
    (_assert_same_colour_map if color else _assert_same_map)(g, o, "final")
    assert g.digest() == g2.digest(), "fast path and general path disagree"
    d, d2 = g.debug(), g2.debug()
    assert d[61] >= n_scans - 3 and d[58] == 0, f"the scans did not take the fast path: {d[58:64]}"
    assert d2[61] == 0
    if mode == "batched":
        assert d[60] < d[59], f"no walk took more than one scan ({d[60]} walks, {d[59]} scans)"


def test_many_handles_keep_their_maps_apart():
    """Three maps fed in turn with pipelined scans (3 x 4 streams on the device's hardware queues, gates spinning on all of
    them): every map equals its own sequential result; hand-over time-outs, if any, only cost time."""
    from ufomap_amd import scans
    maps = [_maps(resolution=0.16) for _ in range(3)]
    for i in range(12):
        for k, (g, o) in enumerate(maps):
            origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose((i + k) % 4), seed=40 * k + i)
            _insert(g, origin, xyz, 10.0, True, True)
            o.insert(origin, xyz, max_range=10.0, discrete=True)
    for k, (g, o) in enumerate(maps):
        g.insertPointCloudWait()
        _assert_same_map(g, o, f"map {k}")



@pytest.mark.parametrize("color", [False, True])
def test_device_cloud_may_be_reused_as_soon_as_the_call_returns(color):
    """ufomap_map_insert_device(async=1): the caller's device buffers are overwritten with garbage the moment each call has
    returned -- while the scan is still in flight, and before the scans that have to be REPEATED (the sensor jumps: the
    predicted ray grid does not fit) are repeated. The map must equal the reference's all the same (include/ufomap_hip.h:
    inputs are consumed before the call returns)."""
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, OccupancyMapColor, scans
    g = (OccupancyMapColor if color else OccupancyMap)(0.16)
    o = OracleMap(0.16, kind="port", color=color)
    base = np.array(scans.lidar_pose(0), dtype=np.float64)
    offs = [(0, 0, 0), (0.05, 0, 0), (0.1, 0.05, 0), (5.0, 3.0, 0.2), (5.1, 3.0, 0.2), (0.1, 0, 0), (0.15, 0, 0), (-4.0, 2.0, 0), (0, 0, 0), (0.02, 0, 0)]
    n = 16 * 512
    d = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    drgb = torch.empty((n, 3), dtype=torch.uint8, device="cuda")
    for i, off in enumerate(offs):
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=512, origin=tuple(base + np.array(off)), seed=300 + i, colored=color)
        d.copy_(torch.from_numpy(xyz))
        if color:
            drgb.copy_(torch.from_numpy(rgb))
        torch.cuda.synchronize()
        g.insert_device(origin, d.data_ptr(), drgb.data_ptr() if color else None, n, 10.0, 0, True, False, 0, True)
        d.fill_(float(1000 + i))  # the buffers are the caller's again
        drgb.fill_(7)
        torch.cuda.synchronize()
        o.insert(origin, xyz, rgb if color else None, max_range=10.0, discrete=True)
    g.insertPointCloudWait()
    gl, ol = g.leaves(True), o.leaves(True)
    assert all(np.array_equal(a, b) for a, b in zip(gl, ol)) and same_dump(g.inner(), o.inner())
    if not color:
        assert g.debug()[63] >= 1, "the jumps should have forced repeats"


@pytest.mark.parametrize("kind", ["pageable", "pageable_colour", "pinned", "inline"])
def test_host_cloud_may_be_reused_as_soon_as_the_call_returns(kind):
    """ufomap_map_insert(async=1) with a full-size HOST cloud (131 072 points: 3 MB, the size from which a helper thread copies the
    cloud into the set's pinned staging buffer while the calling thread enqueues the scan -- round 6; `inline`: the calling thread
    copies, option stage_thread = 0; `pinned`: the caller's own pinned buffer, DMA'd from where it lies and awaited when the call
    ends): the caller's arrays are overwritten the moment each call has returned, scans that have to be repeated (the sensor
    jumps) included. The map must equal the reference's."""
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor, scans
    color = kind == "pageable_colour"
    g = (OccupancyMapColor if color else OccupancyMap)(0.16)
    if kind == "inline":
        g.set_option("stage_thread", 0)
    o = OracleMap(0.16, kind="port", color=color)
    base = np.array(scans.lidar_pose(2), dtype=np.float64)
    offs = [(0, 0, 0), (0.05, 0, 0), (0.1, 0.05, 0), (5.0, 3.0, 0.2), (5.1, 3.0, 0.2), (0.1, 0, 0), (0.15, 0, 0), (0, 0, 0)]
    n = 64 * 2048
    buf = torch.empty((n, 3), dtype=torch.float64).pin_memory().numpy() if kind == "pinned" else np.empty((n, 3), dtype=np.float64)
    cbuf = np.empty((n, 3), dtype=np.uint8)
    for i, off in enumerate(offs):
        origin, xyz, rgb = scans.lidar64(origin=tuple(base + np.array(off)), seed=700 + i, colored=color)
        buf[:] = xyz
        if color:
            cbuf[:] = rgb
        cloud = PointCloudColor(buf, cbuf) if color else PointCloud(buf)
        g.insertPointCloudDiscrete(origin, cloud, 20.0, 0, False, 0, True)
        buf[:] = 1000.0 + i  # the arrays are the caller's again
        cbuf[:] = 7
        o.insert(origin, xyz, rgb if color else None, max_range=20.0, discrete=True)
    g.insertPointCloudWait()
    gl, ol = g.leaves(True), o.leaves(True)
    assert all(np.array_equal(a, b) for a, b in zip(gl, ol)) and same_dump(g.inner(), o.inner())


@pytest.mark.parametrize("color", [False, True])
def test_host_clouds_of_many_sizes_awaited_one_by_one(color):
    """A pageable host cloud whose scan is awaited (nothing in flight when it arrives: the reference's server) reaches HBM in pieces --
    k_stage_copy's workgroups read piece k from the staging buffer while the helper thread copies piece k + 1, round 6 -- from 256 KB on;
    smaller clouds are copied by the calling thread. Sizes around that threshold, odd point counts (24 n and 3 n bytes are then no
    multiples of 16: the last bytes of a piece go one by one), the option's extremes; every scan awaited, the caller's arrays overwritten
    after every call. The map must equal the one the same clouds give from device memory."""
    from ufomap_amd import OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor, scans
    cls = OccupancyMapColor if color else OccupancyMap
    g, ref = cls(0.16), cls(0.16)
    ref.set_option("stage_thread", 0)
    full = 64 * 2048
    sizes = [full, 10923, 10922, 1, 50001, full - 1, 87383, 0, full]
    for i, n in enumerate(sizes):
        g.set_option("stage_pieces", (8, 1, 16, 3)[i % 4])
        origin, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(i % 4), seed=900 + i, colored=color)
        xyz, rgb = xyz[:n], (rgb[:n] if color else None)
        buf, cbuf = xyz.copy(), (rgb.copy() if color else None)
        g.insertPointCloudDiscrete(origin, PointCloudColor(buf, cbuf) if color else PointCloud(buf), 20.0, 0, False, 0, True)
        buf[:] = -5.0
        if color:
            cbuf[:] = 9
        g.insertPointCloudWait()
        ref.insertPointCloudDiscrete(origin, PointCloudColor(xyz, rgb) if color else PointCloud(xyz), 20.0, 0, False, 0, False)
    assert g.digest() == ref.digest()
    assert all(np.array_equal(a, b) for a, b in zip(g.leaves(True), ref.leaves(True)))


_FEW_QUEUES = r"""
import sys, time
import numpy as np
import torch
from ufomap_amd import OccupancyMap, PointCloud, scans
maps = [OccupancyMap(0.16) for _ in range(3)]
refs = [OccupancyMap(0.16) for _ in range(3)]
for r in refs:
    r.set_option("fast", 0)   # general path, synchronous: the checker inside this process
t0 = time.time()
for i in range(16):
    for k in range(3):
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose((i + k) % 4), seed=40 * k + i)
        maps[k].insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0, 0, False, 0, True)
        refs[k].insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0, 0, False, 0, False)
for m in maps:
    m.insertPointCloudWait()
dt = time.time() - t0
ok = all(m.digest() == r.digest() for m, r in zip(maps, refs))
print("RESULT", int(ok), sum(m.debug()[58] for m in maps), sum(m.debug()[61] for m in maps), round(dt, 2))
"""


@pytest.mark.parametrize("queues", ["2", "1"])
def test_gates_with_fewer_hardware_queues_than_streams(queues):
    """GPU_MAX_HW_QUEUES = 2 / 1 and three maps fed in turn: their twelve streams share one or two hardware queues, so a gate
    can sit in front of the very work it waits for. That must only cost time, and a bounded amount of it: a hand-over that
    times out (20 ms) flags its scan, which is repeated, and the handle hands over with events from then on -- every map
    still equals its sequential result."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES=queues, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", _FEW_QUEUES], env=env, capture_output=True, text=True, timeout=280)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    ok, timeouts, fast, secs = line[0].split()[1:]
    assert ok == "1", "a map differs from its sequential result"
    assert int(fast) > 0, "the steady-state path never ran"
    assert float(secs) < 60.0, f"{secs} s for 48 small scans: hand-over time-outs are not bounded ({timeouts} of them)"


# ---- ufomap_map_insert_batch with MORE THAN ONE RANK, through the C ABI, on one GPU -----------------------------------------
# tests/cpp/rccl_shim.cpp stands in for librccl (UFOMAP_RCCL_LIB): two processes on device 0, the all-gather staged through
# POSIX shared memory. What is tested is everything around the collective: the slot/capacity-growth loop of the update-list
# form, the fast-path form (bit grids, one walk for both ranks' scans), the ranks' common ray grid, the collective repeat
# of a step that a rank's scan does not fit, an empty cloud on one rank -- every replica must equal the reference's map
# after the same scans one by one in (step, rank) order.

def _build_shim():
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, out = os.path.join(here, "cpp", "rccl_shim.cpp"), os.path.join(here, "cpp", "librccl_shim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", src, "-o", out, "-lrt"], check=True, capture_output=True)
    return out


def _batch_plan(world, steps):
    """(step, rank) -> (origin, cloud parameters): sensors that drift, one that jumps by metres (the common grid does not fit),
    one empty cloud."""
    from ufomap_amd import scans
    plan = {}
    for i in range(steps):
        for r in range(world):
            base = np.array(scans.lidar_pose(r % 4), dtype=np.float64)
            off = np.array([0.05 * i, -0.03 * i * (r + 1), 0.0])
            if i == 5 and r == world - 1:
                off = off + np.array([7.0, 5.0, 0.2])  # jump: this rank's scan leaves the ranks' common grid
            empty = (i == 3 and r == 0)
            plan[(i, r)] = (tuple(base + off), 700 + 10 * i + r, empty)
    return plan


def _batch_cloud(entry):
    from ufomap_amd import scans
    origin, seed, empty = entry
    o, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=origin, seed=seed)
    return o, (xyz[:0] if empty else xyz)


def _batch_rank(rank, world, steps, shim, id_q, out_q, comm_slot, fail_at=None, device=0, batch_depth=None):
    import os
    if shim:  # (None: the real librccl, one GPU per rank)
        os.environ["UFOMAP_RCCL_LIB"] = shim
    os.environ["UFOMAP_COMM_SLOT"] = str(comm_slot)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch as th
    from ufomap_amd import OccupancyMap
    from ufomap_amd.occupancy_map import Comm
    try:
        if rank == 0:
            uid = Comm.unique_id()
            for _ in range(world - 1):
                id_q.put(uid)
        else:
            uid = id_q.get(timeout=120)
        th.cuda.set_device(device)
        g = OccupancyMap(0.16, device=device)
        g.set_option("async_apply", 1)
        if batch_depth is not None:  # (steps in flight before the oldest is joined: the same on every rank)
            g.set_option("batch_depth", batch_depth)
        comm = Comm(uid, world, rank, device)
        plan = _batch_plan(world, steps)
        keep = []
        for i in range(steps):
            origin, xyz = _batch_cloud(plan[(i, rank)])
            d = th.from_numpy(np.ascontiguousarray(xyz)).cuda(device) if len(xyz) else th.empty(0, dtype=th.float64, device=f"cuda:{device}")
            keep.append(d)
            if fail_at is not None:  # (step, rank): the scan half of that rank's step 'fails' before the collective (option fail_scan)
                g.set_option("fail_scan", int(fail_at == (i, rank)))
            g.insert_batch(comm, origin, d.data_ptr() if len(xyz) else 0, len(xyz), 10.0, 0, True)
        g.insertPointCloudWait()
        out_q.put((rank, g.digest(), comm.counters(), comm.stats(), g.debug()[58:64]))
        comm.close()
    except Exception as e:  # noqa: BLE001
        out_q.put((rank, "error: " + repr(e), None, None, None))


@pytest.mark.parametrize("fail_at,batch_depth", [(None, None), ((7, 1), None), (None, 2), ((7, 1), 5), ((3, 0), 6)])
def test_insert_batch_two_ranks_on_one_gpu(fail_at, batch_depth):
    """fail_at = (step, rank): that rank's scan half fails on the host before the step's all-gather (ADVICE r3 / VERDICT r3 5c) --
    it enters the collective all the same with a flagged empty contribution, nobody hangs, every rank repeats the step in list form
    and the replicas equal the sequential map. batch_depth: how many steps are in flight before the oldest is joined (3 by default,
    2 until round 6) -- a flagged step then has up to five steps enqueued behind it, which stand back and are repeated in order."""
    import torch.multiprocessing as mp
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap
    shim = _build_shim()
    world, steps = 2, 9
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_batch_rank, args=(r, world, steps, shim, id_q, out_q, 4096, fail_at, 0, batch_depth)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r = out_q.get(timeout=300)
        results[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert not isinstance(results[r][1], str), results[r][1]
    # the reference's map after the same scans, one by one, in (step, rank) order
    o = OracleMap(0.16, kind=_kind())
    g1 = OccupancyMap(0.16)  # ... and its digest through the single-GPU path (the digest is computed on the device)
    plan = _batch_plan(world, steps)
    from ufomap_amd import PointCloud
    for i in range(steps):
        for r in range(world):
            origin, xyz = _batch_cloud(plan[(i, r)])
            o.insert(origin, xyz, max_range=10.0, discrete=True)
            g1.insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0)
    _assert_same_map(g1, o, "single-GPU path vs reference")
    want = g1.digest()
    for r in range(world):
        assert results[r][1] == want, f"replica of rank {r} differs from the sequential map"
    c0 = results[0][2]
    assert c0 == results[1][2], "the ranks disagree about which steps took which form"
    assert c0["fast_steps"] >= 4, f"the fast-path form of the step did not run: {c0}"
    assert c0["repeated_steps"] >= (2 if fail_at else 1), f"the jump (and the injected failure) should have forced collective repeats: {c0}"
    assert results[0][3]["regrown"] >= 1, "the update-list slot should have had to grow (4 KiB to start with)"


def test_every_scan_of_a_walk_reports_the_tables_fill():
    """k_ftail wrote the regions' fill (groups claimed, first-region blocks) into the control block of the walk's LAST
    scan only; a scan that was not the last, joined by itself, told the host the table was empty. Walks of four scans, joined one
    scan at a time: the library counts result blocks that report an empty table while the host knows better -- none."""
    from ufomap_amd import OccupancyMap, PointCloud
    g = OccupancyMap(resolution=0.16)
    g.set_option("hold", 4)
    from test_gpu_vol import _wander
    seq = _wander(20, spread=0.3)
    for i, (origin, xyz) in enumerate(seq):
        g.insertPointCloudDiscrete(origin, PointCloud(xyz), 12.0, 0, False, 0, True)
        if i % 5 == 4:
            g.insertPointCloudWait()
    g.insertPointCloudWait()
    d = g.debug()
    assert d[60] < d[59], f"no walk took more than one scan ({d[60]} walks, {d[59]} scans)"
    assert d[46] == 0, f"{d[46]} joined scans reported an empty node table"


def test_insert_batch_two_gpus_real_rccl():
    """The same two-rank sequence over the REAL librccl, one GPU per rank (ncclAllGather across processes over xGMI / PCIe): runs
    wherever two GPUs are visible -- the driver's 8-GPU node --, skipped on a one-GPU box (there the shim above stands in). Every
    replica must equal the map of the same scans one by one in (step, rank) order."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one-GPU boxes run test_insert_batch_two_ranks_on_one_gpu through the RCCL stand-in)")
    import torch.multiprocessing as mp
    from ufomap_amd import OccupancyMap, PointCloud
    world, steps = 2, 9
    ctx = mp.get_context("spawn")
    id_q, out_q = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_batch_rank, args=(r, world, steps, None, id_q, out_q, 4096, None, r)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        r = out_q.get(timeout=600)
        results[r[0]] = r
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert not isinstance(results[r][1], str), results[r][1]
    g1 = OccupancyMap(0.16)
    plan = _batch_plan(world, steps)
    for i in range(steps):
        for r in range(world):
            origin, xyz = _batch_cloud(plan[(i, r)])
            g1.insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0)
    want = g1.digest()
    for r in range(world):
        assert results[r][1] == want, f"replica of rank {r} differs from the sequential map"
    assert results[0][2] == results[1][2] and results[0][2]["fast_steps"] >= 4


def test_insert_batch_simple_ray_casting_and_early_stopping():
    """ufomap_map_insert_batch_ex: the two remaining arguments of insertPointCloudDiscrete (occupancy_map_base.h:340-344) in a batch
    step -- fixed-step casting and early stopping take the update-list form; world 1 through the real RCCL, against the reference."""
    import torch
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, Comm, scans
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind=_kind())
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        for s in range(6):
            origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s % 3), seed=100 + s, beams=16, azimuths=512)
            simple, early = (s % 3 == 1), (3 if s % 3 == 2 else 0)
            d = torch.from_numpy(xyz).cuda()
            g.insert_batch(comm, origin, d.data_ptr(), xyz.shape[0], 12.0, 0, True, None, simple, early)
            o.insert(origin, xyz, max_range=12.0, discrete=True, simple_ray_casting=simple, early_stopping=early)
            torch.cuda.synchronize()
        g.insertPointCloudWait()
        _assert_same_map(g, o, "batch steps with simple_ray_casting / early_stopping")
    finally:
        g.insertPointCloudWait()
        comm.close()


def test_gate_timeouts_are_survived():
    """a stream hand-over that gives up (option gate_us at its minimum, 100 us, against host clouds whose upload alone takes
    longer) flags its scan -- which leaves the map alone and is repeated once the streams have drained -- and the handle hands over
    with events from then on: the map equals the reference's whether or not a gate timed out (the counter says how many did)."""
    from ufomap_amd import PointCloud, scans
    g, o = _maps(kind=_kind(), resolution=0.16)
    g.set_option("gate_us", 100)
    n_to = 0
    for i in range(10):
        origin, xyz, _ = scans.lidar64(origin=tuple(np.array(scans.lidar_pose(1)) + [0.05 * i, 0.0, 0.0]), seed=40 + i)
        g.insertPointCloudDiscrete(origin, PointCloud(xyz.copy()), 20.0, 0, False, 0, True)  # a pageable host cloud: memcpy + H2D before k_fhits
        o.insert(origin, xyz, max_range=20.0, discrete=True)
        if i == 4:
            g.insertPointCloudWait()
            _assert_same_map(g, o, "after scan 4")
    g.insertPointCloudWait()
    _assert_same_map(g, o, "final")
    n_to = g.debug()[58]
    assert n_to >= 0
    print("gate timeouts:", n_to)


@pytest.mark.parametrize("seeds", ["3 120 11", "4 150 13", "2 150 23", "2 150 40", "2 150 55"])
def test_random_call_sequences_against_the_checker(seeds):
    """scripts/dev/fuzz_api.py: random sequences of calls on one map -- host / device / PointCloud2 clouds, synchronous and asynchronous,
    insert depths, continuous and discrete, robot clearing, batch steps over a one-rank communicator that comes and goes, byte streams,
    clear -- compared with the CPU checker after every few calls, several maps one after the other in one process (recycled device
    memory). The seeds are the ones that found round 6's four bugs in the seams between the kinds of calls: a scan behind a batch step
    that stood back applied before the step's repeat; a batch step as a map's first steady-state update on an unzeroed record array; a
    batch step repeated from inside a PointCloud2 call reading its cloud through that call's record layout; the word behind the
    all-gather polled before it was zeroed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev", "fuzz_api.py")] + seeds.split(), capture_output=True, text=True, timeout=900, cwd=root)
    assert 0 == pr.returncode, pr.stdout[-3000:] + pr.stderr[-1500:]


@pytest.mark.parametrize("mode", [{"FUZZ_WORLD": "3", "FUZZ_CHG": "box", "FUZZ_EXT": "1"}, {"FUZZ_COLOR": "1"}, {"FUZZ_CHG": "codes", "FUZZ_EXT": "1"}])
def test_random_call_sequences_other_modes(mode):
    """The same with this process playing rank 0 of three (every batch step applies the scan three times; the min / max change box of every
    replica grows by every rank's scan), on a colour map, and with the per-code change set compared as a set of nodes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _build_shim()
    pr = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev", "fuzz_api.py"), "3", "80", "1200"], capture_output=True, text=True, timeout=900, cwd=root,
                        env=dict(os.environ, **mode))
    assert 0 == pr.returncode, pr.stdout[-3000:] + pr.stderr[-1500:]


def test_single_scans_after_batch_steps_and_a_destroyed_communicator():
    """A hand-over set that held a batch step kept the step's communicator until round 6: taking a scan of ufomap_map_insert next, it
    predicted the RANKS' grid through that pointer when the scan was joined -- freed memory once ufomap_comm_destroy had run (bench.py's
    batch_step_n1 leg followed by its host legs: one run in eight died of a corrupted heap) -- and skipped the map's own prediction.
    Batch steps over every set, the communicator destroyed, a long row of single scans (the sensor jumps twice: repeats), more batch
    steps on a second communicator, single scans again: the map equals the reference's, and the single scans keep taking the steady-state
    path (their ray grid is predicted from their own boxes again)."""
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, PointCloud, scans
    from ufomap_amd.occupancy_map import Comm
    g = OccupancyMap(0.16)
    o = OracleMap(0.16, kind=_kind())
    keep = []
    k = 0

    def cloud(i, jump=0.0):
        return scans.lidar64(beams=16, azimuths=512, origin=tuple(np.array(scans.lidar_pose(1)) + [0.03 * i + jump, 0.0, 0.0]), seed=300 + i)[:2]

    for phase in range(2):
        comm = Comm(Comm.unique_id(), 1, 0, 0)
        g.set_option("async_apply", 1)
        for _ in range(12):  # (more steps than there are hand-over sets: every set has carried one)
            origin, xyz = cloud(k)
            d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
            keep.append(d)
            g.insert_batch(comm, origin, d.data_ptr(), len(xyz), 10.0, 0, True)
            o.insert(origin, xyz, max_range=10.0, discrete=True)
            k += 1
        g.insertPointCloudWait()
        g.set_option("async_apply", 0)
        assert comm.counters()["fast_steps"] >= 6
        comm.close()
        del comm
        junk = [bytearray(200) for _ in range(2000)]  # (the freed communicator's memory is likely to be handed out again)
        fast0 = g.debug()[61]
        for i in range(20):
            origin, xyz = cloud(k, jump=4.0 if i in (7, 8, 9) else 0.0)
            g.insertPointCloudDiscrete(origin, PointCloud(xyz), 10.0, 0, False, 0, True)
            o.insert(origin, xyz, max_range=10.0, discrete=True)
            k += 1
        g.insertPointCloudWait()
        del junk
        assert g.debug()[61] - fast0 >= 14, "the single scans after the batch steps should take the steady-state path"
    _assert_same_map(g, o, "batch steps, destroyed communicators and single scans on one map")


def test_insert_batch_at_insert_depth():
    """Batch steps at insert depth > 0 (occupancy_map_base.h:378-386, 1085-1120; the way the reference's README tells RGB-D users to run
    a fine map): the update-list form, the ranks' lists applied one by one in rank order -- world 1 through the real RCCL, steps at
    depth 0 (bit-grid form once the common grid is known), 1, 2 and 3 mixed, against the reference scan by scan."""
    import torch
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, Comm, scans
    g, o = OccupancyMap(0.08), OracleMap(0.08, kind=_kind())
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        for s, depth in enumerate([0, 0, 2, 0, 1, 3, 0, 2, 2, 0]):
            origin, xyz, _ = scans.lidar64(origin=tuple(np.array(scans.lidar_pose(1)) + [0.07 * s, 0.03 * s, 0.0]), seed=300 + s, beams=16, azimuths=512)
            d = torch.from_numpy(xyz).cuda()
            g.insert_batch(comm, origin, d.data_ptr(), xyz.shape[0], 9.0, depth, True)
            o.insert(origin, xyz, max_range=9.0, depth=depth, discrete=True)
            torch.cuda.synchronize()
            if s in (2, 5):
                g.insertPointCloudWait()
                _assert_same_map(g, o, f"after step {s} (insert depth {depth})")
        g.insertPointCloudWait()
        _assert_same_map(g, o, "batch steps at insert depths 0 .. 3")
    finally:
        g.insertPointCloudWait()
        comm.close()


def test_insert_batch_at_insert_depth_three_ranks():
    """... and with three ranks played by one process (tests/cpp/rccl_shim.cpp: every slot of the all-gather a copy of the caller's):
    every odd step at insert depth 2 -- the lists of the three ranks applied one after the other, hits then misses each, as the
    reference does scan after scan -- every even step at depth 0."""
    import torch.multiprocessing as mp
    from ufomap_amd import scans
    from oracle import OracleMap
    import golden_util
    world, steps, depth = 3, 6, 2
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    p = ctx.Process(target=_replicated_world, args=(world, steps, out_q, _build_shim(), depth))
    p.start()
    dig, counters = out_q.get(timeout=300)
    p.join(timeout=60)
    assert not isinstance(dig, str), dig
    o = OracleMap(0.16, kind=_kind())
    for i in range(steps):
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=tuple(np.array(scans.lidar_pose(1)) + [0.04 * i, 0.0, 0.0]), seed=70 + i)
        for _ in range(world):
            o.insert(origin, xyz, max_range=10.0, depth=depth if i % 2 else 0, discrete=True)
    assert tuple(dig) == tuple(golden_util.dump_digest(o.leaves(True), o.inner())), "the replica differs from the sequential map"


def _replicated_world(world, steps, out_q, shim, depth=0):
    """One process plays all `world` ranks (tests/cpp/rccl_shim.cpp, UFOMAP_SHIM_REPLICATE: every slot of an all-gather is a copy of the
    caller's): a step applies this rank's scan `world` times, in order."""
    import os
    os.environ["UFOMAP_RCCL_LIB"] = shim
    os.environ["UFOMAP_SHIM_REPLICATE"] = "1"
    import torch as th
    from ufomap_amd import OccupancyMap, scans
    from ufomap_amd.occupancy_map import Comm
    try:
        g = OccupancyMap(0.16)
        g.set_option("async_apply", 1)
        comm = Comm(Comm.unique_id(), world, 0, 0)
        keep = []
        for i in range(steps):
            origin, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=tuple(np.array(scans.lidar_pose(1)) + [0.04 * i, 0.0, 0.0]), seed=70 + i)
            d = th.from_numpy(np.ascontiguousarray(xyz)).cuda()
            keep.append(d)
            g.insert_batch(comm, origin, d.data_ptr(), len(xyz), 10.0, depth if i % 2 else 0, True)
        g.insertPointCloudWait()
        out_q.put((g.digest(), comm.counters()))
        comm.close()
    except Exception as e:  # noqa: BLE001
        out_q.put(("error: " + repr(e), None))


@pytest.mark.parametrize("world", [16, 17])
def test_insert_batch_form_by_world_size(world):
    """one walk takes at most UFO_BATCH_MAX = 16 scans, so a communicator of more ranks must not take the bit-grid
    form (several walks per step would each look at their own chunk's flags only). 16 ranks: bit-grid steps; 17: the update-list form
    throughout -- either way the replica equals the reference integrating every step's scan `world` times."""
    import torch.multiprocessing as mp
    from ufomap_amd import scans
    steps = 6
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    p = ctx.Process(target=_replicated_world, args=(world, steps, out_q, _build_shim()))
    p.start()
    dig, counters = out_q.get(timeout=300)
    p.join(timeout=60)
    assert not isinstance(dig, str), dig
    from oracle import OracleMap
    import golden_util
    o = OracleMap(0.16, kind=_kind())
    for i in range(steps):
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=256, origin=tuple(np.array(scans.lidar_pose(1)) + [0.04 * i, 0.0, 0.0]), seed=70 + i)
        for _ in range(world):
            o.insert(origin, xyz, max_range=10.0, discrete=True)
    assert tuple(dig) == tuple(golden_util.dump_digest(o.leaves(True), o.inner())), "the replica differs from the sequential map"
    if world > 16:
        assert counters["fast_steps"] == 0, f"a step of {world} ranks took the bit-grid form: {counters}"
    else:
        assert counters["fast_steps"] >= 2, f"no bit-grid step with {world} ranks: {counters}"
