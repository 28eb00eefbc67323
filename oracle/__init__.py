"""CPU checkers for the integration path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package (see ``oracle/oracle_abi.h``).  Two interchangeable back-ends export the same C ABI:

* ``kind="port"``       ``oracle/libufo_oracle.so`` -- our CPU restatement (``ufo_oracle.cpp``)
* ``kind="reference"``  ``oracle/_ref/libufo_ref.so`` -- the unmodified reference compiled from
  ``/root/reference`` by ``oracle/Makefile`` (prebuilt file is used where the reference tree is
  absent, e.g. on the GPU box)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {
    "port": os.path.join(_HERE, "libufo_oracle.so"),
    "reference": os.path.join(_HERE, "_ref", "libufo_ref.so"),
}
_LIBS: dict = {}


class RunawayRay(RuntimeError):
    """A ray left the map cube after clipping (end point a few ulp outside): the reference walks
    ~2^31 cells on such input; the port refuses (rc=-4) before touching the map."""


def build(kind: str = "port", quiet: bool = True) -> bool:
    """(Re)build a checker with oracle/Makefile. Returns True if the library exists afterwards."""
    target = "port" if kind == "port" else "ref"
    if kind == "reference" and not os.path.isdir("/root/reference/ufomap/include"):
        return os.path.exists(_PATHS[kind])
    r = subprocess.run(["make", "-C", _HERE, target], capture_output=quiet, text=True)
    if r.returncode != 0 and not quiet:
        print(r.stdout, r.stderr)
    return os.path.exists(_PATHS[kind])


def available(kind: str) -> bool:
    return os.path.exists(_PATHS[kind])


def _load(kind: str):
    if kind in _LIBS:
        return _LIBS[kind]
    path = _PATHS[kind]
    if not os.path.exists(path):
        build(kind)
    lib = C.CDLL(path)
    vp, u64p, u8p, f32p, f64p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.ufo_oracle_create.restype = vp
    lib.ufo_oracle_create.argtypes = [C.c_double, C.c_uint, C.c_int] + [C.c_double] * 6 + [C.c_int]
    lib.ufo_oracle_destroy.argtypes = [vp]
    lib.ufo_oracle_insert.restype = C.c_int
    lib.ufo_oracle_insert.argtypes = [vp, f64p, f64p, u8p, C.c_size_t, C.c_double, C.c_uint, C.c_int, C.c_int, C.c_uint]
    lib.ufo_oracle_export_leaves.restype = C.c_size_t
    lib.ufo_oracle_export_leaves.argtypes = [vp, C.c_int, u64p, u8p, f32p, u8p, C.c_size_t]
    lib.ufo_oracle_export_inner.restype = C.c_size_t
    lib.ufo_oracle_export_inner.argtypes = [vp, u64p, u8p, f32p, u8p, u8p, C.c_size_t]
    lib.ufo_oracle_write.restype = C.c_size_t
    lib.ufo_oracle_write.argtypes = [vp, u8p, C.c_size_t]
    lib.ufo_oracle_minmax_change.restype = C.c_int
    lib.ufo_oracle_minmax_change.argtypes = [vp, f64p, f64p]
    lib.ufo_oracle_last_hits.restype = C.c_size_t
    lib.ufo_oracle_last_hits.argtypes = [vp, u64p, C.c_size_t]
    lib.ufo_oracle_last_rays.restype = C.c_size_t
    lib.ufo_oracle_last_rays.argtypes = [vp, f64p, C.c_size_t]
    lib.ufo_oracle_last_misses.restype = C.c_size_t
    lib.ufo_oracle_last_misses.argtypes = [vp, u64p, C.c_size_t]
    lib.ufo_oracle_last_steps.restype = C.c_uint64
    lib.ufo_oracle_last_steps.argtypes = [vp]
    lib.ufo_oracle_last_oob.restype = C.c_uint64
    lib.ufo_oracle_last_oob.argtypes = [vp]
    lib.ufo_oracle_query.restype = None
    lib.ufo_oracle_query.argtypes = [vp, C.POINTER(C.c_double), C.c_size_t, C.c_uint, C.POINTER(C.c_float), u8p]
    lib.ufo_oracle_set_value_volume.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_uint]
    lib.ufo_oracle_clamping_thres.restype = None
    lib.ufo_oracle_clamping_thres.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ufo_oracle_ingest.restype = C.c_size_t
    lib.ufo_oracle_ingest.argtypes = [u8p, C.c_size_t, C.c_uint32] + [C.c_int] * 6 + [C.POINTER(C.c_double)] * 3 + [u8p]
    lib.ufo_oracle_iterate.restype = C.c_size_t
    lib.ufo_oracle_iterate.argtypes = [vp, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, u64p, u8p, f32p, u8p, u8p, C.c_size_t]
    lib.ufo_oracle_enable_change_detection.argtypes = [vp, C.c_int]
    lib.ufo_oracle_reset_change_detection.argtypes = [vp]
    lib.ufo_oracle_changes.restype = C.c_size_t
    lib.ufo_oracle_changes.argtypes = [vp, u64p, u8p, C.c_size_t]
    lib.ufo_oracle_enable_minmax_change_detection.argtypes = [vp, C.c_int]
    lib.ufo_oracle_write_ex.restype = C.c_size_t
    lib.ufo_oracle_write_ex.argtypes = [vp, f64p, f64p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_longlong)]
    lib.ufo_oracle_read.argtypes = [vp, u8p, C.c_size_t]
    lib.ufo_oracle_read_data.argtypes = [vp, u8p, C.c_size_t, f64p, f64p, C.c_double, C.c_uint, C.c_int, C.c_int]
    lib.ufo_oracle_get_sensor_model.argtypes = [vp, f64p]
    lib.ufo_oracle_set_model_value.argtypes = [vp, C.c_int, C.c_double]
    lib.ufo_oracle_set_occupied_free_thres.argtypes = [vp, C.c_double, C.c_double]
    lib.ufo_oracle_clear_to.argtypes = [vp, C.c_double, C.c_uint]
    lib.ufo_oracle_set_value_volume_ch.argtypes = [vp, f64p, f64p, C.c_double, C.c_uint]
    lib.ufo_oracle_kind.restype = C.c_char_p
    _LIBS[kind] = lib
    return lib


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def ingest(data, step, off_xyz, off_rgb, rot_wxyz, trans, kind="port"):
    """rosToUfo + PointCloud::transform of the reference on a raw PointCloud2-style byte buffer.
    data: uint8 array of n*step bytes; off_xyz = (x, y, z) byte offsets of float32 fields; off_rgb = (r, g, b)
    byte offsets or None. Returns (xyz float64 [k, 3], rgb uint8 [k, 3]) of the k points without NaN."""
    lib = _load(kind)
    data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    n = data.size // step
    xyz = np.empty((max(n, 1), 3), np.float64)
    rgb = np.empty((max(n, 1), 3), np.uint8)
    q = np.ascontiguousarray(rot_wxyz, dtype=np.float64)
    t = np.ascontiguousarray(trans, dtype=np.float64)
    orgb = off_rgb if off_rgb is not None else (-1, -1, -1)
    k = lib.ufo_oracle_ingest(_ptr(data, C.c_uint8), n, step, *[int(o) for o in off_xyz], *[int(o) for o in orgb],
                              _ptr(q, C.c_double), _ptr(t, C.c_double), _ptr(xyz, C.c_double), _ptr(rgb, C.c_uint8))
    return xyz[:k].copy(), rgb[:k].copy()


class OracleMap:
    """ctypes view of one CPU checker map (argument meaning = OccupancyMapBase ctor, OMB:859-862)."""

    def __init__(self, resolution, depth_levels=16, automatic_pruning=True, occupied_thres=0.5,
                 free_thres=0.5, prob_hit=0.7, prob_miss=0.4, clamping_thres_min=0.1192,
                 clamping_thres_max=0.971, color=False, kind="port"):
        self.lib = _load(kind)
        self.kind = kind
        self.color = bool(color)
        self.h = self.lib.ufo_oracle_create(resolution, depth_levels, int(automatic_pruning),
                                            occupied_thres, free_thres, prob_hit, prob_miss,
                                            clamping_thres_min, clamping_thres_max, int(color))
        if not self.h:
            raise ValueError("oracle: invalid constructor arguments")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ufo_oracle_destroy(self.h)
            self.h = None

    def insert(self, origin, xyz, rgb=None, max_range=-1.0, depth=0, discrete=False,
               simple_ray_casting=False, early_stopping=0):
        origin = np.ascontiguousarray(origin, dtype=np.float64)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
            assert rgb.shape[0] == xyz.shape[0]
        rc = self.lib.ufo_oracle_insert(self.h, _ptr(origin, C.c_double), _ptr(xyz, C.c_double),
                                        _ptr(rgb, C.c_uint8), xyz.shape[0], float(max_range),
                                        int(depth), int(discrete), int(simple_ray_casting),
                                        int(early_stopping))
        if rc == -4:
            raise RunawayRay("input outside the parity contract (runaway ray), map unchanged")
        if rc != 0:
            raise RuntimeError(f"oracle insert failed rc={rc}")

    def leaves(self, include_unknown=False):
        n = self.lib.ufo_oracle_export_leaves(self.h, int(include_unknown), None, None, None, None, 0)
        codes = np.empty(n, np.uint64)
        depths = np.empty(n, np.uint8)
        occ = np.empty(n, np.float32)
        rgb = np.zeros((n, 3), np.uint8)
        self.lib.ufo_oracle_export_leaves(self.h, int(include_unknown), _ptr(codes, C.c_uint64),
                                          _ptr(depths, C.c_uint8), _ptr(occ, C.c_float),
                                          _ptr(rgb, C.c_uint8), n)
        return codes, depths, occ, rgb

    def inner(self):
        n = self.lib.ufo_oracle_export_inner(self.h, None, None, None, None, None, 0)
        codes = np.empty(n, np.uint64)
        depths = np.empty(n, np.uint8)
        occ = np.empty(n, np.float32)
        flags = np.empty(n, np.uint8)
        rgb = np.zeros((n, 3), np.uint8)
        self.lib.ufo_oracle_export_inner(self.h, _ptr(codes, C.c_uint64), _ptr(depths, C.c_uint8),
                                         _ptr(occ, C.c_float), _ptr(flags, C.c_uint8),
                                         _ptr(rgb, C.c_uint8), n)
        return codes, depths, occ, flags, rgb

    def write(self):
        """The map as the reference's .ufo byte stream (Octree::write, uncompressed)."""
        n = self.lib.ufo_oracle_write(self.h, None, 0)
        buf = np.empty(n, np.uint8)
        self.lib.ufo_oracle_write(self.h, _ptr(buf, C.c_uint8), n)
        return buf.tobytes()

    def query(self, xyz, depth=0):
        """Per coordinate: (log-odds of the node Octree::getNode returns, state bits: 1 occupied, 2 free,
        4 unknown, 8 containsFree, 16 containsUnknown)."""
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        lo = np.empty(xyz.shape[0], np.float32)
        st = np.empty(xyz.shape[0], np.uint8)
        self.lib.ufo_oracle_query(self.h, _ptr(xyz, C.c_double), xyz.shape[0], int(depth), _ptr(lo, C.c_float), _ptr(st, C.c_uint8))
        return lo, st

    def setValueVolume(self, aabb_min, aabb_max, occupancy_value, min_depth=0):
        """OccupancyMapBase::setValueVolume(AABB(min, max), occupancy_value, min_depth) (robot clearing)."""
        mn = np.ascontiguousarray(aabb_min, np.float64)
        mx = np.ascontiguousarray(aabb_max, np.float64)
        self.lib.ufo_oracle_set_value_volume(self.h, _ptr(mn, C.c_double), _ptr(mx, C.c_double), float(occupancy_value), int(min_depth))

    def clamping_thres(self):
        a, b = C.c_double(), C.c_double()
        self.lib.ufo_oracle_clamping_thres(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def minmax_change(self):
        mn = np.empty(3, np.float64)
        mx = np.empty(3, np.float64)
        self.lib.ufo_oracle_minmax_change(self.h, _ptr(mn, C.c_double), _ptr(mx, C.c_double))
        return mn, mx


    # ---- round-2 rows: the reference build only (oracle_abi.h) ------------------------------------------------
    @staticmethod
    def _bv(aabb):
        if aabb is None:
            return None, None, None
        c = np.ascontiguousarray(aabb[0], np.float64)
        h = np.ascontiguousarray(aabb[1], np.float64)
        return (c, h), _ptr(c, C.c_double), _ptr(h, C.c_double)

    def _need(self, ok):
        if not ok:
            raise NotImplementedError("provided by the reference build only (oracle/_ref/libufo_ref.so)")

    def iterate(self, aabb=None, occupied_space=True, free_space=True, unknown_space=False, contains=False, min_depth=0, only_leaves=True):
        keep, pc, ph = self._bv(aabb)
        a = (int(occupied_space), int(free_space), int(unknown_space), int(contains), int(min_depth), int(only_leaves))
        n = self.lib.ufo_oracle_iterate(self.h, pc, ph, *a, None, None, None, None, None, 0)
        self._need(n != C.c_size_t(-1).value)
        codes, depths, occ = np.empty(n, np.uint64), np.empty(n, np.uint8), np.empty(n, np.float32)
        rgb, flags = np.zeros((n, 3), np.uint8), np.empty(n, np.uint8)
        self.lib.ufo_oracle_iterate(self.h, pc, ph, *a, _ptr(codes, C.c_uint64), _ptr(depths, C.c_uint8), _ptr(occ, C.c_float),
                                    _ptr(rgb, C.c_uint8), _ptr(flags, C.c_uint8), n)
        return codes, depths, occ, rgb, flags

    def enableChangeDetection(self, enable=True):
        self._need(self.lib.ufo_oracle_enable_change_detection(self.h, int(enable)) == 0)

    def resetChangeDetection(self):
        self._need(self.lib.ufo_oracle_reset_change_detection(self.h) == 0)

    def changes(self):
        n = self.lib.ufo_oracle_changes(self.h, None, None, 0)
        self._need(n != C.c_size_t(-1).value)
        codes, depths = np.empty(n, np.uint64), np.empty(n, np.uint8)
        self.lib.ufo_oracle_changes(self.h, _ptr(codes, C.c_uint64), _ptr(depths, C.c_uint8), n)
        return codes, depths

    def enableMinMaxChangeDetection(self, enable=True):
        self._need(self.lib.ufo_oracle_enable_minmax_change_detection(self.h, int(enable)) == 0)

    def write_ex(self, aabb=None, compress=False, min_depth=0, compression_acceleration_level=1, compression_level=0, header=True):
        keep, pc, ph = self._bv(aabb)
        us = C.c_longlong(-1)
        a = (int(compress), int(min_depth), int(compression_acceleration_level), int(compression_level), int(header))
        n = self.lib.ufo_oracle_write_ex(self.h, pc, ph, *a, None, 0, C.byref(us))
        self._need(n != C.c_size_t(-1).value)
        buf = np.empty(max(n, 1), np.uint8)
        self.lib.ufo_oracle_write_ex(self.h, pc, ph, *a, _ptr(buf, C.c_uint8), n, C.byref(us))
        return buf[:n].tobytes(), int(us.value)

    def read(self, data):
        b = np.frombuffer(data, np.uint8)
        self._need(self.lib.ufo_oracle_read(self.h, _ptr(b, C.c_uint8), b.size) == 0)

    def readData(self, data, resolution, depth_levels, uncompressed_data_size=1, compressed=False, aabb=None):
        b = np.frombuffer(data, np.uint8)
        keep, pc, ph = self._bv(aabb)
        self._need(self.lib.ufo_oracle_read_data(self.h, _ptr(b, C.c_uint8) if b.size else None, b.size, pc, ph, float(resolution), int(depth_levels),
                                                 int(uncompressed_data_size), int(compressed)) == 0)

    def sensor_model(self):
        out = np.zeros(6, np.float64)
        self._need(self.lib.ufo_oracle_get_sensor_model(self.h, _ptr(out, C.c_double)) == 0)
        return tuple(float(v) for v in out)

    def set_model_value(self, which, p):
        self._need(self.lib.ufo_oracle_set_model_value(self.h, int(which), float(p)) == 0)

    def setOccupiedFreeThres(self, occupied_thres, free_thres):
        self._need(self.lib.ufo_oracle_set_occupied_free_thres(self.h, float(occupied_thres), float(free_thres)) == 0)

    def clear_to(self, resolution, depth_levels):
        self._need(self.lib.ufo_oracle_clear_to(self.h, float(resolution), int(depth_levels)) == 0)

    def setValueVolumeAABB(self, center, half_size, occupancy_value, min_depth=0):
        c, h = np.ascontiguousarray(center, np.float64), np.ascontiguousarray(half_size, np.float64)
        self._need(self.lib.ufo_oracle_set_value_volume_ch(self.h, _ptr(c, C.c_double), _ptr(h, C.c_double), float(occupancy_value), int(min_depth)) == 0)

    # -- stage-level outputs of the last insert (port only) ---------------------------------
    def _stage(self, fn, dtype, width=1):
        n = fn(self.h, None, 0)
        if n == C.c_size_t(-1).value:
            raise NotImplementedError("stage outputs are only provided by the port")
        out = np.empty((n, width) if width > 1 else n, dtype)
        ct = C.c_uint64 if dtype == np.uint64 else C.c_double
        fn(self.h, _ptr(out, ct), n)
        return out

    def last_hits(self):
        return self._stage(self.lib.ufo_oracle_last_hits, np.uint64)

    def last_misses(self):
        return self._stage(self.lib.ufo_oracle_last_misses, np.uint64)

    def last_rays(self):
        return self._stage(self.lib.ufo_oracle_last_rays, np.float64, 3)

    def last_steps(self):
        return int(self.lib.ufo_oracle_last_steps(self.h))

    def last_oob(self):
        """Keys of the last insert outside [0, 2^L) (port only; such input is outside the parity contract)."""
        return int(self.lib.ufo_oracle_last_oob(self.h))
