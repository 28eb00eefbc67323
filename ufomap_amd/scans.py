"""Deterministic synthetic scans -- inputs for parity tests and bench.py.

The reference ships no datasets; BASELINE.json's configs are quoted on these generators:

* ``lidar64``  64 beams x 2048 azimuths = 131 072 points, sensor inside a 30 x 24 x 5 m box,
  +-1 cm range noise (configs C1, C2, C4, C5).
* ``rgbd``     640 x 480 pinhole depth image, 307 200 points at 1..3 m, +-1 mm noise (config C3).

Randomness is splitmix64 (portable, vectorised): draw k of a stream seeded with ``seed`` uses state
``seed + (k+1) * 0x9E3779B97F4A7C15``.  Points are float64 xyz; colours uint8 in 1..255 per channel
(never (0,0,0), which the reference treats as "unset": map/color.h:85).
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, count: int) -> np.ndarray:
    """First ``count`` outputs of splitmix64 seeded with ``seed`` (uint64)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _draws(seed: int, n: int, colored: bool):
    """Per point: one draw for noise, then three for r,g,b when coloured (draw order of 8d)."""
    per = 4 if colored else 1
    z = splitmix64(seed, n * per).reshape(n, per)
    u = (z[:, 0] >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    rgb = None
    if colored:
        rgb = (np.uint64(1) + z[:, 1:4] % np.uint64(255)).astype(np.uint8)
    return u, rgb


LIDAR_ORIGIN = (0.1, 0.2, 1.7)
LIDAR_BOX = ((-15.0, 15.0), (-12.0, 12.0), (0.0, 5.0))


def lidar64(origin=LIDAR_ORIGIN, seed: int = 42, colored: bool = False, beams: int = 64,
            azimuths: int = 2048):
    """64-beam LiDAR scan of the inside of LIDAR_BOX. Returns (origin[3], xyz[N,3], rgb[N,3]|None)."""
    o = np.asarray(origin, dtype=np.float64)
    el = np.deg2rad(-24.8 + (2.0 - (-24.8)) * np.arange(beams, dtype=np.float64) / max(beams - 1, 1))
    az = 2.0 * np.pi * np.arange(azimuths, dtype=np.float64) / azimuths
    el, az = np.meshgrid(el, az, indexing="ij")  # beam-major, azimuth-minor
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1).reshape(-1, 3)
    n = d.shape[0]
    t = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax in range(3):
            for plane in LIDAR_BOX[ax]:
                tt = (plane - o[ax]) / d[:, ax]
                tt = np.where(tt > 0, tt, np.inf)
                t = np.minimum(t, tt)
    u, rgb = _draws(seed, n, colored)
    t = t + (2.0 * u - 1.0) * 0.01
    xyz = o[None, :] + t[:, None] * d
    return o, np.ascontiguousarray(xyz), rgb


def lidar_pose(s: int):
    """Pose s of the 8-scan batch of config C4; seed is 100+s."""
    return (-10.5 + 3.0 * s, 0.2 + (s % 3) - 1.0, 1.7)


def rgbd(origin=LIDAR_ORIGIN, seed: int = 42, colored: bool = False, width: int = 640,
         height: int = 480):
    """640x480 RGB-D frame looking along +x (config C3). Returns (origin, xyz, rgb|None)."""
    o = np.asarray(origin, dtype=np.float64)
    v, uu = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    fx = 525.0 * width / 640.0
    x = (uu - (width - 1) / 2.0) / fx
    y = (v - (height - 1) / 2.0) / fx
    n = width * height
    u, rgb = _draws(seed, n, colored)
    z = 2.0 + np.sin(0.01 * uu * 640.0 / width) * np.cos(0.013 * v * 480.0 / height)
    z = z.reshape(-1) + (2.0 * u - 1.0) * 0.001
    xyz = np.stack([o[0] + z, o[1] - x.reshape(-1) * z, o[2] - y.reshape(-1) * z], axis=-1)
    return o, np.ascontiguousarray(xyz), rgb


def random_cloud(n: int, seed: int, extent: float = 8.0, origin=(0.3, -0.2, 0.4), colored: bool = False):
    """Uniform random points in a cube of half-width ``extent`` (edge-case fuzzing)."""
    z = splitmix64(seed, n * 3).reshape(n, 3)
    u = (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    xyz = (2.0 * u - 1.0) * extent
    rgb = None
    if colored:
        zz = splitmix64(seed ^ 0x5DEECE66D, n * 3).reshape(n, 3)
        rgb = (np.uint64(1) + zz % np.uint64(255)).astype(np.uint8)
    return np.asarray(origin, dtype=np.float64), np.ascontiguousarray(xyz), rgb
