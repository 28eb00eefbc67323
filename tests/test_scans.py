"""Host logic: the deterministic scan generators of SURVEY.md 8(d)."""
import numpy as np

from ufomap_amd import scans


def test_splitmix64_known_values():
    # splitmix64 with seed 0: first outputs of the published reference implementation
    z = scans.splitmix64(0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_lidar64_shape_and_determinism():
    o, xyz, rgb = scans.lidar64()
    assert xyz.shape == (131072, 3) and xyz.dtype == np.float64 and rgb is None
    o2, xyz2, _ = scans.lidar64()
    assert np.array_equal(xyz, xyz2)
    # every point lies on the box (x in [-15,15], y in [-12,12], z in [0,5]) up to the +-1 cm noise
    assert np.all(np.abs(xyz[:, 0]) <= 15.02) and np.all(np.abs(xyz[:, 1]) <= 12.02)
    assert xyz[:, 2].min() > -0.02 and xyz[:, 2].max() < 5.02
    rng = np.linalg.norm(xyz - o, axis=1)
    assert 1.0 < rng.min() and rng.max() < 20.0


def test_colours_never_unset():
    _, _, rgb = scans.lidar64(beams=4, azimuths=64, colored=True)
    assert rgb.dtype == np.uint8 and rgb.min() >= 1  # (0,0,0) means "unset" in the reference (color.h:85)


def test_batch_poses_differ():
    assert scans.lidar_pose(0) != scans.lidar_pose(1)
    _, a, _ = scans.lidar64(origin=scans.lidar_pose(0), seed=100, beams=4, azimuths=64)
    _, b, _ = scans.lidar64(origin=scans.lidar_pose(1), seed=101, beams=4, azimuths=64)
    assert not np.array_equal(a, b)


def test_rgbd_shape():
    o, xyz, rgb = scans.rgbd(colored=True)
    assert xyz.shape == (307200, 3) and rgb.shape == (307200, 3)
    d = xyz[:, 0] - o[0]
    assert 0.99 < d.min() and d.max() < 3.01
