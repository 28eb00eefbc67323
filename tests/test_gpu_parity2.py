"""Parity tests added in round 2 (run with -m gpu on an MI355X): the configurations round 1 only checked through
size-independent properties, the unmodified reference as the direct checker, digests of maps too large to dump,
colour through the pipelined / speculative paths, pinned host clouds."""
import numpy as np
import pytest

import golden_util
from conftest import same_dump

pytestmark = pytest.mark.gpu


def _kind():
    """The unmodified reference build when it travelled to this box (oracle/_ref/libufo_ref.so), else our port."""
    import oracle
    return "reference" if oracle.available("reference") else "port"


def _maps(color=False, kind="port", **params):
    from oracle import OracleMap
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    g = (OccupancyMapColor if color else OccupancyMap)(**params)
    o = OracleMap(kind=kind, color=color, **params)
    return g, o


def _gpu_insert(g, origin, xyz, rgb=None, max_range=-1.0, depth=0, discrete=False, simple_ray_casting=False, async_=False):
    from ufomap_amd import PointCloud, PointCloudColor
    cloud = PointCloudColor(xyz, rgb) if rgb is not None else PointCloud(xyz)
    (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, cloud, max_range, depth, simple_ray_casting, 0, async_)


def _assert_same_map(g, o, what=""):
    gl, ol = g.leaves(True), o.leaves(True)
    assert len(gl[0]) == len(ol[0]), f"{what}: leaf count {len(gl[0])} vs oracle {len(ol[0])}"
    assert np.array_equal(gl[0], ol[0]) and np.array_equal(gl[1], ol[1]), f"{what}: leaf codes/depths differ"
    assert np.array_equal(gl[2], ol[2]), f"{what}: log-odds differ"
    assert np.array_equal(gl[3], ol[3]), f"{what}: colours differ"
    assert same_dump(g.inner(), o.inner()), f"{what}: inner-node dump differs"
    assert g.write() == o.write(), f"{what}: map byte stream differs"


@pytest.mark.parametrize("color", [False, True])
def test_device_digest_equals_host_twin(color):
    """ufomap_map_digest (one pass on the device, no export) == golden_util.dump_digest of the exported dumps."""
    from ufomap_amd import scans
    g, _ = _maps(color=color, resolution=0.16)
    assert g.digest() == golden_util.dump_digest(g.leaves(True), g.inner())  # fresh map: the root alone
    for s in range(3):
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose(s), seed=s, colored=color)
        _gpu_insert(g, origin, xyz, rgb, max_range=10.0, discrete=True)
        assert g.digest(True) == golden_util.dump_digest(g.leaves(True), g.inner())
        assert g.digest(False)[:3] == golden_util.dump_digest(g.leaves(False), g.inner())[:3]


@pytest.mark.parametrize("name", sorted(golden_util.digests()))
def test_digest_fixtures_from_the_reference(name):
    """Full BASELINE configurations against fingerprints of the UNMODIFIED reference's dumps
    (tests/golden/make_digests.py) after every scan -- including config C3 at insert depth 0 at full size
    (3.4e8 leaves; 85-170 s and 20 GB per scan on the CPU), which no test could compare before."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_digests import make_scan
    from ufomap_amd import OccupancyMap, OccupancyMapColor
    fx = golden_util.digests()[name]
    params = dict(fx["params"])
    color = params.pop("color", False)
    m = (OccupancyMapColor if color else OccupancyMap)(**params)
    for i, ((gen, gkw, ikw), st) in enumerate(zip(fx["scans"], fx["steps"])):
        origin, xyz, rgb = make_scan(gen, gkw)
        _gpu_insert(m, origin, xyz, rgb, **ikw)
        assert m.digest() == tuple(int(v) for v in st["digest"]), f"{name}: scan {i}: digest differs from the reference's"


def test_c3_depth0_reduced_frame_leaf_for_leaf():
    """BASELINE config C3 (2 mm leaf, 5 m) at insert depth 0 -- the bandwidth-bound configuration: huge ray grids
    (k_dda on > 10-bit local coordinates, direct global marking), repeated table growth -- on a 160x120 frame
    (3e7 DDA steps, 1.1e8 leaves), leaf for leaf and inner node for inner node against the CPU checker, twice
    (fresh map, then the same scan into the grown map)."""
    from ufomap_amd import scans
    g, o = _maps(kind=_kind(), resolution=0.002)
    origin, xyz, _ = scans.rgbd(width=160, height=120)
    for i in range(2):
        _gpu_insert(g, origin, xyz, max_range=5.0, discrete=True)
        o.insert(origin, xyz, max_range=5.0, discrete=True)
        gl, ol = g.leaves(True), o.leaves(True)
        assert same_dump(gl, ol), f"scan {i}: leaves differ"
        assert same_dump(g.inner(), o.inner()), f"scan {i}: inner nodes differ"
        assert g.digest() == golden_util.dump_digest(ol, o.inner())
    assert same_dump(g.minmax_change(), o.minmax_change())


def test_c3_coloured_reduced_frame_leaf_for_leaf():
    """Round 5: the same reduced frame WITH COLOURS into an OccupancyMapColor (the volume path's colour instance, k_tile<true, VOL>):
    leaf for leaf -- codes, depths, log-odds, colours -- and inner node for inner node against the CPU checker, fresh and warm."""
    from ufomap_amd import scans
    g, o = _maps(color=True, kind=_kind(), resolution=0.002)
    origin, xyz, rgb = scans.rgbd(width=160, height=120, colored=True)
    for i in range(2):
        _gpu_insert(g, origin, xyz, rgb, max_range=5.0, discrete=True)
        o.insert(origin, xyz, rgb, max_range=5.0, discrete=True)
        gl, ol = g.leaves(True), o.leaves(True)
        assert same_dump(gl, ol), f"scan {i}: leaves (with colours) differ"
        assert same_dump(g.inner(), o.inner()), f"scan {i}: inner nodes differ"
    assert g.debug()[50] == 2, "the coloured frame did not take the volume path"


@pytest.mark.parametrize("cfg", ["C1", "C2", "C5"])
def test_full_configs_against_the_unmodified_reference(cfg):
    """C1 / C2 / C5 at full size with the reference build itself as the checker (one hop less than the port)."""
    import oracle
    from ufomap_amd import scans
    if not oracle.available("reference"):
        pytest.skip("oracle/_ref/libufo_ref.so not present")
    color = cfg == "C5"
    g, o = _maps(color=color, kind="reference", resolution=0.08 if color else 0.16)
    for s in range(2):
        origin, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, colored=color)
        _gpu_insert(g, origin, xyz, rgb, max_range=20.0, discrete=cfg != "C1")
        o.insert(origin, xyz, rgb, max_range=20.0, discrete=cfg != "C1")
        _assert_same_map(g, o, f"{cfg} scan {s}")
    assert same_dump(g.minmax_change(), o.minmax_change())


@pytest.mark.parametrize("seed", [11, 12])
def test_random_operation_sequence_colour(seed):
    """The interleaved-caller test on an OccupancyMapColor: sync and pipelined coloured inserts (host clouds, device
    clouds, fused PointCloud2 ingest with packed rgb), insert depth 0-2, robot clearing, point queries, a sensor that
    drifts and jumps (predicted grids fit, then do not) -- colour through the pipelined / speculative paths."""
    import torch
    import oracle
    from ufomap_amd import scans
    rng = np.random.default_rng(seed)
    g, o = _maps(color=True, kind=_kind(), resolution=0.16)
    pos = np.array(scans.lidar_pose(1), dtype=np.float64)
    keep = []
    for step in range(24):
        pos = pos + (rng.uniform(-3, 3, 3) * [1, 1, 0.1] if rng.random() < 0.15 else rng.uniform(-0.08, 0.08, 3) * [1, 1, 0.2])
        origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=tuple(pos), seed=1000 * seed + step, colored=True)
        mr = float(rng.choice([6.0, 9.0, 12.0]))
        op = rng.choice(["insert", "insert", "async", "async", "async_host", "pc2", "clear", "depth"])
        if op == "insert":
            _gpu_insert(g, origin, xyz, rgb, max_range=mr, discrete=True)
            o.insert(origin, xyz, rgb, max_range=mr, discrete=True)
        elif op == "async":
            d, dc = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
            keep += [d, dc]
            g.insert_device(origin, d.data_ptr(), dc.data_ptr(), xyz.shape[0], mr, 0, discrete=True, async_=True)
            o.insert(origin, xyz, rgb, max_range=mr, discrete=True)
        elif op == "async_host":
            _gpu_insert(g, origin, xyz, rgb, max_range=mr, discrete=True, async_=True)
            xyz[:] = 0  # the caller's buffers are free as soon as the call has returned
            rgb[:] = 0
            origin, xyz, rgb = scans.lidar64(beams=16, azimuths=256, origin=tuple(pos), seed=1000 * seed + step, colored=True)
            o.insert(origin, xyz, rgb, max_range=mr, discrete=True)
        elif op == "depth":
            dep = int(rng.integers(1, 3))
            _gpu_insert(g, origin, xyz, rgb, max_range=mr, depth=dep, discrete=True)
            o.insert(origin, xyz, rgb, max_range=mr, depth=dep, discrete=True)
        elif op == "pc2":
            f = (xyz - origin).astype(np.float32)
            f[::31, int(rng.integers(0, 3))] = np.nan
            buf = np.zeros((f.shape[0], 16), np.uint8)
            buf[:, 0:12] = f.view(np.uint8).reshape(-1, 12)
            buf[:, 12], buf[:, 13], buf[:, 14] = rgb[:, 2], rgb[:, 1], rgb[:, 0]
            q = np.array([1.0, 0.0, 0.0, 0.0])
            g.insertPointCloud2(origin, q, buf, 16, (0, 4, 8), (14, 13, 12), max_range=mr, async_=bool(rng.integers(0, 2)))
            cx, cc = oracle.ingest(buf, 16, (0, 4, 8), (14, 13, 12), q, origin, "port")
            o.insert(origin, cx, cc, max_range=mr, discrete=True)
        elif op == "clear":
            md = int(rng.integers(0, 3))
            ext = rng.uniform(0.3, 0.9, 3)
            g.setValueVolume(pos - ext, pos + ext, g.getClampingThresMin(), md)
            o.setValueVolume(pos - ext, pos + ext, o.clamping_thres()[0], md)
        if step % 8 == 7:
            g.insertPointCloudWait()
            keep.clear()
            assert same_dump(g.leaves(True), o.leaves(True)), f"seed {seed} step {step} ({op}): leaves differ"
            assert same_dump(g.inner(), o.inner()), f"seed {seed} step {step} ({op}): inner nodes differ"
            qs = np.concatenate([xyz[::11], rng.uniform(-12, 12, (500, 3)) + pos])
            d = int(rng.integers(0, 4))
            a, b = g.query(qs, d), o.query(qs, d)
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    g.insertPointCloudWait()
    _assert_same_map(g, o, f"seed {seed} final")


def test_pinned_and_pageable_host_clouds_pipelined():
    """ufomap_map_insert with async=true: a pageable cloud goes through the hand-over set's pinned staging buffer (no
    synchronisation, the H2D copy of scan i+1 overlaps scan i), a cloud in caller-owned pinned memory is copied by
    DMA straight from it. Either way the caller may overwrite its buffer as soon as the call has returned."""
    import torch
    from ufomap_amd import scans
    g, o = _maps(resolution=0.16)
    pinned = torch.empty((64 * 1024, 3), dtype=torch.float64).pin_memory()
    for s in range(10):
        origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s % 8), seed=100 + s, beams=32, azimuths=1024)
        o.insert(origin, xyz, max_range=20.0, discrete=True)
        if s & 1:
            buf = pinned[: xyz.shape[0]].numpy()
            buf[:] = xyz
        else:
            buf = xyz.copy()
        _gpu_insert(g, origin, buf, max_range=20.0, discrete=True, async_=True)
        buf[:] = np.nan  # scribble over the caller's buffer right after the call
    g.insertPointCloudWait()
    _assert_same_map(g, o, "host clouds")


@pytest.mark.parametrize("mode", ["small_limit", "near_wrap"])
def test_phase_tags_restart(mode):
    """The per-phase tags in the table (24 bits in tmax, 22 in lu_fl) are cleared and the numbering restarts before
    the shortest can wrap (phaseGuard): same maps as the reference across restarts, pruning history included -- with
    the restart forced every few scans, and with the counter started just below the real limit."""
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMap, PointCloud, scans
    kind = "reference" if available("reference") else "port"
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind=kind)
    if mode == "small_limit":
        g.set_option("phase_limit", 7)
    else:
        g.set_option("scan_id", (1 << 22) - (1 << 12) - 5)
    for i in range(14):
        p = [0, 0, 1, 1, 0, 2, 2, 2, 0, 1, 0, 0, 2, 1][i]
        origin, xyz, _ = scans.lidar64(beams=16, azimuths=512, origin=scans.lidar_pose(p), seed=100 + p)
        discrete = bool(i % 3)
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), 12.0, 0, False, 0, bool(i % 2))
        o.insert(origin, xyz, max_range=12.0, discrete=discrete)
        if i % 4 == 3:
            g.insertPointCloudWait()
            assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner()), f"scan {i}"
    g.insertPointCloudWait()
    assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())
    assert g.write() == o.write()
    resets = g.debug()[51]
    assert resets >= (3 if mode == "small_limit" else 1), resets


def test_insert_batch_c_abi_world1():
    """ufomap_map_insert_batch: scan -> RCCL all-gather of update-list slots -> apply, all inside the C library (the
    communicator comes from ufomap_comm_create; no torch.distributed anywhere). One rank: the code path of the 8-GPU
    run minus the peers. The slot capacity starts small here so that the all-ranks-grow-alike rule is exercised too."""
    import torch
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMap, Comm, scans
    kind = "reference" if available("reference") else "port"
    g, o = OccupancyMap(0.16), OracleMap(0.16, kind=kind)
    g.set_option("async_apply", 1)
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        for s in range(5):
            origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(s % 3), seed=100 + s % 3, beams=32, azimuths=1024)
            d = torch.from_numpy(xyz).cuda()
            g.insert_batch(comm, origin, d.data_ptr(), xyz.shape[0], 20.0, 0, bool(s % 2))
            o.insert(origin, xyz, max_range=20.0, discrete=bool(s % 2))
            torch.cuda.synchronize()
        g.insertPointCloudWait()
        assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())
        assert g.write() == o.write()
        st = comm.stats()
        assert st["world"] == 1 and st["slot_bytes"] >= 1 << 20
    finally:
        g.insertPointCloudWait()
        comm.close()


@pytest.mark.parametrize("case", ["depth0_16cm", "depth0_8cm_colour", "depth2_4cm", "continuous_16cm"])
def test_sparse_ray_cell_set_beyond_scratch_limit(case):
    """A scan whose bounding box needs more scratch than the limit allows goes through the sparse set of ray cells
    (Grid::layout 2) instead of failing: the reference's CodeMap has no size bound (code.h:568-785). Forced here by an
    option; the set starts at 64 Ki slots and doubles until the scan fits (the 8 cm case needs several rounds)."""
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor, scans
    kind = "reference" if available("reference") else "port"
    res, depth, color, discrete, full = {"depth0_16cm": (0.16, 0, False, True, False), "depth0_8cm_colour": (0.08, 0, True, True, True),
                                         "depth2_4cm": (0.04, 2, False, True, False), "continuous_16cm": (0.16, 0, False, False, False)}[case]
    g = (OccupancyMapColor if color else OccupancyMap)(res)
    o = OracleMap(res, kind=kind, color=color)
    g.set_option("sparse_set", 1)
    g.set_option("spec", 0)
    kw = {} if full else dict(beams=32, azimuths=1024)
    for s in range(3):
        origin, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, colored=color, **kw)
        cloud = PointCloudColor(xyz, rgb) if color else PointCloud(xyz)
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, cloud, 20.0, depth, False, 0, False)
        o.insert(origin, xyz, rgb, max_range=20.0, depth=depth, discrete=discrete)
        if kind == "port":
            assert g.last_counts()["steps"] == o.last_steps()
    assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())
    assert g.write() == o.write()


def test_far_returns_take_the_sparse_set_by_themselves():
    """Three returns 1.5 - 3 km away in an otherwise small 2 cm scan: the scan's box would need ~10^14 bytes as a dense
    grid (far beyond the default 16 GB scratch limit), the rays touch ~400 k cells. No option set: the sparse set is what
    the library falls back to on its own, and the map equals the reference's."""
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMap, PointCloud, scans
    kind = "reference" if available("reference") else "port"
    g, o = OccupancyMap(0.02, depth_levels=20), OracleMap(0.02, depth_levels=20, kind=kind)  # (the reference smashes its stack at depth_levels 21)
    origin, xyz, _ = scans.lidar64(beams=8, azimuths=256)
    far = np.array([[1500.3, 2.1, 40.7], [-30.2, 2950.9, -12.4], [800.5, -700.25, 300.125]])
    xyz = np.ascontiguousarray(np.concatenate([xyz * 0.2, far]))
    for discrete in (True, False):
        (g.insertPointCloudDiscrete if discrete else g.insertPointCloud)(origin, PointCloud(xyz), -1.0, 0, False, 0, False)
        o.insert(origin, xyz, max_range=-1.0, discrete=discrete)
    assert g.last_counts()["steps"] > 300000
    assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())


def test_colour_update_lists_batch_equals_sequential():
    """Colour maps through the split path: each scan's update list carries a colour section (the colour of the first
    point of every hit voxel), the lists of a batch are applied with one walk of the tree, and the replica equals
    the reference's OccupancyMapColor after insertPointCloudDiscrete of the scans one by one (occupancy_map_color.h:
    177-267) -- values, colours, inner summaries, pruning. Moving sensor, overlapping voxels, three batches of three."""
    import torch
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMapColor, scans
    kind = "reference" if available("reference") else "port"
    g, o = OccupancyMapColor(0.16), OracleMap(0.16, kind=kind, color=True)
    scratch = OccupancyMapColor(0.16)  # a second handle does the ray casting (as another GPU would)
    for batch in range(3):
        bufs, infos = [], []
        for k in range(3):
            p = (batch + 2 * k) % 4
            origin, xyz, rgb = scans.lidar64(beams=32, azimuths=1024, origin=scans.lidar_pose(p), seed=100 + p + 10 * batch, colored=True)
            d, dc = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
            caster = scratch if k % 2 else g
            info = caster.scan_keys(origin, d.data_ptr(), xyz.shape[0], 15.0, 0, True, d_rgb_ptr=dc.data_ptr())
            assert info.reserved & 2 and info.list_bytes == (info.n_hit + info.n_miss) * 16 + info.n_hit * 32
            buf = torch.empty(info.list_bytes, dtype=torch.uint8, device="cuda")
            caster.get_keys(buf.data_ptr(), buf.numel() // 16, info)
            bufs.append(buf)
            infos.append(info)
            o.insert(origin, xyz, rgb, max_range=15.0, discrete=True)
        g.apply_keys_batch([b.data_ptr() for b in bufs], infos)
        g.insertPointCloudWait()
        assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner()), f"batch {batch}"
    assert g.write() == o.write()


def test_insert_batch_c_abi_world1_colour():
    """ufomap_map_insert_batch on an OccupancyMapColor: the colour sections travel through the RCCL all-gather."""
    import torch
    from oracle import OracleMap, available
    from ufomap_amd import OccupancyMapColor, Comm, scans
    kind = "reference" if available("reference") else "port"
    g, o = OccupancyMapColor(0.16), OracleMap(0.16, kind=kind, color=True)
    comm = Comm(Comm.unique_id(), 1, 0, 0)
    try:
        for s in range(4):
            origin, xyz, rgb = scans.lidar64(origin=scans.lidar_pose(s % 3), seed=100 + s % 3, beams=32, azimuths=1024, colored=True)
            d, dc = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
            g.insert_batch(comm, origin, d.data_ptr(), xyz.shape[0], 20.0, 0, True, d_rgb_ptr=dc.data_ptr())
            o.insert(origin, xyz, rgb, max_range=20.0, discrete=True)
            torch.cuda.synchronize()
        g.insertPointCloudWait()
        assert g.digest() == golden_util.dump_digest(o.leaves(True), o.inner())
    finally:
        g.insertPointCloudWait()
        comm.close()
