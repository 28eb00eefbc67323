"""Host-side mirror of the reference's map API for the integration path (Python flavour).

Mirrors ``ufo::map::OccupancyMap`` / ``ufo::map::OccupancyMapColor`` of the reference
(ufomap/include/ufo/map/occupancy_map.h:55-85, occupancy_map_color.h:56-98): same method names,
argument order, defaults and error behaviour for the hot path

    insertPointCloud(sensor_origin, cloud, max_range=-1, depth=0, simple_ray_casting=False,
                     early_stopping=0, async_=False)                    (occupancy_map_base.h:270-273)
    insertPointCloudDiscrete(...same tail...)                           (occupancy_map_base.h:340-344)
    insertPointCloudDone() / insertPointCloudWait()                     (occupancy_map_base.h:430-443)

Everything forwards to the C ABI of ``include/ufomap_hip.h``; the map lives in HBM.  The C++ twin
is ``include/ufomap_amd/occupancy_map.hpp``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class PointCloud:
    """``ufo::map::PointCloud`` (point_cloud.h:277): N x 3 float64 points."""

    def __init__(self, xyz=None):
        self.xyz = np.zeros((0, 3), np.float64) if xyz is None else np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        self.rgb = None

    def size(self):
        return self.xyz.shape[0]

    __len__ = size


class PointCloudColor(PointCloud):
    """``ufo::map::PointCloudColor`` (point_cloud.h:278): points + 3 x uint8 colour."""

    def __init__(self, xyz=None, rgb=None):
        super().__init__(xyz)
        self.rgb = np.zeros((self.size(), 3), np.uint8) if rgb is None else np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
        if self.rgb.shape[0] != self.size():
            raise ValueError("xyz and rgb differ in length")


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class OccupancyMapBase:
    _color = False

    def __init__(self, resolution, depth_levels=16, automatic_pruning=True, occupied_thres=0.5, free_thres=0.5,
                 prob_hit=0.7, prob_miss=0.4, clamping_thres_min=0.1192, clamping_thres_max=0.971, device=0):
        self._lib = capi.load()
        self._h = self._lib.ufomap_map_create(resolution, depth_levels, int(automatic_pruning), occupied_thres, free_thres,
                                              prob_hit, prob_miss, clamping_thres_min, clamping_thres_max, int(self._color), device)
        if not self._h:
            msg = self._lib.ufomap_last_error().decode()
            # the reference throws std::invalid_argument for bad depth_levels (octree.h:931-935)
            raise ValueError(msg) if "depth_levels" in msg or "resolution" in msg else capi.UfomapError(capi.ERR_DEVICE, msg)
        self.resolution = resolution
        self.depth_levels = depth_levels

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ufomap_map_destroy(h)

    # ---- tree type strings (occupancy_map.h:84, occupancy_map_color.h:84-87) -------------------
    def getTreeType(self):
        return "occupancy_map_color" if self._color else "occupancy_map"

    # ---- integration -----------------------------------------------------------------------------
    def _insert(self, sensor_origin, cloud, max_range, depth, discrete, simple_ray_casting, early_stopping, async_):
        if isinstance(cloud, PointCloud):
            xyz, rgb = cloud.xyz, cloud.rgb
        else:
            xyz, rgb = np.ascontiguousarray(cloud, np.float64).reshape(-1, 3), None
        o = np.ascontiguousarray(sensor_origin, np.float64)
        capi.check(self._lib.ufomap_map_insert(self._h, _p(o, C.c_double), xyz.ctypes.data if xyz.size else None,
                                               rgb.ctypes.data if rgb is not None and rgb.size else None, xyz.shape[0],
                                               float(max_range), int(depth), int(discrete), int(simple_ray_casting),
                                               int(early_stopping), int(async_)))

    def insertPointCloud(self, sensor_origin, cloud, max_range=-1.0, depth=0, simple_ray_casting=False, early_stopping=0,
                         async_=False):
        self._insert(sensor_origin, cloud, max_range, depth, False, simple_ray_casting, early_stopping, async_)

    def insertPointCloudDiscrete(self, sensor_origin, cloud, max_range=-1.0, depth=0, simple_ray_casting=False,
                                 early_stopping=0, async_=False):
        self._insert(sensor_origin, cloud, max_range, depth, True, simple_ray_casting, early_stopping, async_)

    def insert_device(self, sensor_origin, d_xyz_ptr, d_rgb_ptr, n, max_range=-1.0, depth=0, discrete=True,
                      simple_ray_casting=False, early_stopping=0, async_=False):
        """Same as the two calls above for a cloud already resident in HBM (raw device pointers)."""
        # (the ctypes pointer of an origin array is kept: building one costs 2.3 us, a twentieth of a pipelined scan's period -- the
        # host's time per call is what bounds the steady-state path)
        o = sensor_origin
        if not (isinstance(o, np.ndarray) and o.dtype == np.float64 and o.flags.c_contiguous):
            o = np.ascontiguousarray(o, np.float64)
        cache = self.__dict__.setdefault("_origin_ptrs", {})
        ent = cache.get(id(o))
        if ent is None or ent[0] is not o:
            if len(cache) > 64:
                cache.clear()
            ent = cache[id(o)] = (o, _p(o, C.c_double))
        capi.check(self._lib.ufomap_map_insert_device(self._h, ent[1], d_xyz_ptr, d_rgb_ptr, n, float(max_range),
                                                      int(depth), int(discrete), int(simple_ray_casting), int(early_stopping),
                                                      int(async_)))

    def insertPointCloud2(self, translation, rotation_wxyz, data, point_step, off_xyz, off_rgb=None, max_range=-1.0, depth=0,
                          discrete=True, simple_ray_casting=False, early_stopping=0, async_=False, n_points=None):
        """rosToUfo + cloud.transform(pose) + insertPointCloudDiscrete(pose.translation(), cloud, ...) of the
        reference's server (ufomap_mapping/src/server.cpp:114-120) on the raw records of a PointCloud2:
        ``data`` = uint8 numpy array (host) or an int device pointer (then pass ``n_points``); float32 x, y, z
        at byte offsets ``off_xyz``; bytes r, g, b at ``off_rgb`` (None: no colour)."""
        t = np.ascontiguousarray(translation, np.float64)
        q = np.ascontiguousarray(rotation_wxyz, np.float64)
        orgb = tuple(int(o) for o in off_rgb) if off_rgb is not None else (-1, -1, -1)
        if isinstance(data, (int, np.integer)):
            ptr, on_dev, n = int(data), 1, int(n_points)
        else:
            data = np.ascontiguousarray(data, np.uint8).reshape(-1)
            ptr, on_dev, n = data.ctypes.data if data.size else None, 0, data.size // int(point_step)
        capi.check(self._lib.ufomap_map_insert_pointcloud2(self._h, _p(t, C.c_double), _p(q, C.c_double), ptr, on_dev, n, int(point_step),
                                                           int(off_xyz[0]), int(off_xyz[1]), int(off_xyz[2]), *orgb, float(max_range),
                                                           int(depth), int(discrete), int(simple_ray_casting), int(early_stopping),
                                                           int(async_)))

    # ---- point queries (occupancy_map_base.h:599-728), batched ---------------------------------------
    OCCUPIED, FREE, UNKNOWN, CONTAINS_FREE, CONTAINS_UNKNOWN = 1, 2, 4, 8, 16

    def query(self, xyz, depth=0):
        """Per coordinate: (log-odds, state bits) of the node the reference's getNode returns; see
        ``ufomap_map_query`` in include/ufomap_hip.h. ``xyz``: [n, 3] float64 (host)."""
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        n = xyz.shape[0]
        lo = np.empty(n, np.float32)
        st = np.empty(n, np.uint8)
        capi.check(self._lib.ufomap_map_query(self._h, xyz.ctypes.data if n else None, 0, n, int(depth), _p(lo, C.c_float),
                                              _p(st, C.c_uint8)))
        return lo, st

    def getState(self, coord, depth=0):
        return int(self.query(coord, depth)[1][0]) & 7

    def isOccupied(self, coord, depth=0):
        return self.getState(coord, depth) == self.OCCUPIED

    def isFree(self, coord, depth=0):
        return self.getState(coord, depth) == self.FREE

    def isUnknown(self, coord, depth=0):
        return self.getState(coord, depth) == self.UNKNOWN

    def containsFree(self, coord, depth=0):
        return bool(self.query(coord, depth)[1][0] & self.CONTAINS_FREE)

    def containsUnknown(self, coord, depth=0):
        return bool(self.query(coord, depth)[1][0] & self.CONTAINS_UNKNOWN)

    def setValueVolume(self, aabb_min, aabb_max, occupancy_value, min_depth=0):
        """OccupancyMapBase::setValueVolume(ufo::geometry::AABB(min, max), occupancy_value, min_depth)
        (occupancy_map_base.h:492-518): the server's robot clearing (ufomap_mapping/src/server.cpp:152-155)."""
        mn = np.ascontiguousarray(aabb_min, np.float64)
        mx = np.ascontiguousarray(aabb_max, np.float64)
        capi.check(self._lib.ufomap_map_set_value_volume(self._h, _p(mn, C.c_double), _p(mx, C.c_double), float(occupancy_value),
                                                         int(min_depth)))

    def getClampingThresMin(self):
        a = np.zeros(2, np.float64)
        capi.check(self._lib.ufomap_map_clamping_thres(self._h, _p(a[0:1], C.c_double), _p(a[1:2], C.c_double)))
        return float(a[0])

    def getClampingThresMax(self):
        a = np.zeros(2, np.float64)
        capi.check(self._lib.ufomap_map_clamping_thres(self._h, _p(a[0:1], C.c_double), _p(a[1:2], C.c_double)))
        return float(a[1])


    # ---- what the reference's callers use around the hot path (round 2; include/ufomap_hip.h) ---------------------
    @staticmethod
    def _bv(aabb):
        """aabb = None or (centre[3], half_size[3]) -- the members of ufo::geometry::AABB."""
        if aabb is None:
            return None, None, None
        c = np.ascontiguousarray(aabb[0], np.float64)
        h = np.ascontiguousarray(aabb[1], np.float64)
        return (c, h), _p(c, C.c_double), _p(h, C.c_double)

    def iterate(self, aabb=None, occupied_space=True, free_space=True, unknown_space=False, contains=False, min_depth=0, only_leaves=True):
        """``beginLeaves`` (only_leaves) / ``beginTree`` (occupancy_map_base.h:93-165) run to the end, on the device:
        (codes >> 3*depth, depths, log-odds, rgb, flags) in the iterator's order."""
        keep, pc, ph = self._bv(aabb)
        a = (int(occupied_space), int(free_space), int(unknown_space), int(contains), int(min_depth), int(only_leaves))
        n = self._lib.ufomap_map_iterate(self._h, pc, ph, *a, None, None, None, None, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        codes, depths, occ = np.empty(n, np.uint64), np.empty(n, np.uint8), np.empty(n, np.float32)
        rgb, flags = np.zeros((n, 3), np.uint8), np.empty(n, np.uint8)
        if n:
            got = self._lib.ufomap_map_iterate(self._h, pc, ph, *a, _p(codes, C.c_uint64), _p(depths, C.c_uint8), _p(occ, C.c_float),
                                               _p(rgb, C.c_uint8), _p(flags, C.c_uint8), n)
            assert got == n
        return codes, depths, occ, rgb, flags

    def enableChangeDetection(self, enable=True):
        capi.check(self._lib.ufomap_map_enable_change_detection(self._h, int(enable)))

    def resetChangeDetection(self):
        capi.check(self._lib.ufomap_map_reset_change_detection(self._h))

    def changes(self):
        """The change set (occupancy_map_base.h:779-791) as (codes >> 3*depth, depths), sorted by (depth, code)."""
        n = self._lib.ufomap_map_changes(self._h, None, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        codes, depths = np.empty(n, np.uint64), np.empty(n, np.uint8)
        if n:
            self._lib.ufomap_map_changes(self._h, _p(codes, C.c_uint64), _p(depths, C.c_uint8), n)
        return codes, depths

    def enableMinMaxChangeDetection(self, enable=True):
        capi.check(self._lib.ufomap_map_enable_minmax_change_detection(self._h, int(enable)))

    def write_ex(self, aabb=None, compress=False, min_depth=0, compression_acceleration_level=1, compression_level=0, header=True):
        """``Octree::write`` (header) / ``writeData`` with all arguments (octree.h:779-917). Returns (bytes, uncompressed size)."""
        keep, pc, ph = self._bv(aabb)
        us = C.c_longlong(-1)
        a = (int(compress), int(min_depth), int(compression_acceleration_level), int(compression_level), int(header))
        # one serialisation when the stream fits the buffer kept from the last call (the C call returns the size it needs)
        buf = getattr(self, "_wbuf", None)
        if buf is None:
            buf = self._wbuf = np.empty(1 << 16, np.uint8)
        n = self._lib.ufomap_map_write_ex(self._h, pc, ph, *a, _p(buf, C.c_uint8), buf.size, C.byref(us))
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        if n > buf.size:
            buf = np.empty(n + n // 2, np.uint8)
            n = self._lib.ufomap_map_write_ex(self._h, pc, ph, *a, _p(buf, C.c_uint8), buf.size, C.byref(us))
            if n == C.c_size_t(-1).value or n > buf.size:
                capi.check(-2)
            # (a stream of tens of megabytes is not kept for the object's lifetime)
            self._wbuf = buf if buf.size <= (8 << 20) else None
        return buf[:n].tobytes(), int(us.value)

    def read(self, data):
        """``Octree::read(std::istream&)``: header + node stream as ``write`` produces them."""
        b = np.frombuffer(data, np.uint8)
        res, lv = C.c_double(), C.c_uint()
        capi.check(self._lib.ufomap_map_read(self._h, _p(b, C.c_uint8), b.size, C.byref(res), C.byref(lv)))
        self.resolution, self.depth_levels = res.value, lv.value

    def readData(self, data, resolution, depth_levels, uncompressed_data_size=1, compressed=False, aabb=None):
        """``Octree::readData`` (octree.h:737-777): a UFOMap message's node stream merged into the map."""
        b = np.frombuffer(data, np.uint8)
        keep, pc, ph = self._bv(aabb)
        capi.check(self._lib.ufomap_map_read_data(self._h, _p(b, C.c_uint8) if b.size else None, b.size, pc, ph, float(resolution), int(depth_levels),
                                                  int(uncompressed_data_size), int(compressed)))
        self.resolution, self.depth_levels = resolution, depth_levels

    def sensor_model(self):
        """(occupied_thres, free_thres, prob_hit, prob_miss, clamping_thres_min, clamping_thres_max) as the reference's getters return them."""
        out = np.zeros(6, np.float64)
        capi.check(self._lib.ufomap_map_get_sensor_model(self._h, _p(out, C.c_double)))
        return tuple(float(v) for v in out)

    def setProbHit(self, p):
        capi.check(self._lib.ufomap_map_set_model_value(self._h, 2, float(p)))

    def setProbMiss(self, p):
        capi.check(self._lib.ufomap_map_set_model_value(self._h, 3, float(p)))

    def setClampingThresMin(self, p):
        capi.check(self._lib.ufomap_map_set_model_value(self._h, 4, float(p)))

    def setClampingThresMax(self, p):
        capi.check(self._lib.ufomap_map_set_model_value(self._h, 5, float(p)))

    def setOccupiedFreeThres(self, occupied_thres, free_thres):
        capi.check(self._lib.ufomap_map_set_occupied_free_thres(self._h, float(occupied_thres), float(free_thres)))

    def clear_to(self, resolution, depth_levels):
        """``Octree::clear(new_resolution, new_depth_levels)`` (octree.h:544-575)."""
        capi.check(self._lib.ufomap_map_clear_to(self._h, float(resolution), int(depth_levels)))
        self.resolution, self.depth_levels = resolution, depth_levels

    def setValueVolumeAABB(self, center, half_size, occupancy_value, min_depth=0):
        c, h = np.ascontiguousarray(center, np.float64), np.ascontiguousarray(half_size, np.float64)
        capi.check(self._lib.ufomap_map_set_value_volume_ch(self._h, _p(c, C.c_double), _p(h, C.c_double), float(occupancy_value), int(min_depth)))

    # ---- multi-GPU batched scans: the path split at its exchange point (include/ufomap_hip.h) ------
    ENTRY_BYTES = 16

    def scan_keys(self, sensor_origin, d_xyz_ptr, n, max_range=-1.0, depth=0, discrete=True, simple_ray_casting=False, d_rgb_ptr=None):
        """Ray-cast one scan WITHOUT touching the map; returns the header of its update list (colour maps: pass the
        points' colours, the list then carries a colour section: ``info.list_bytes``)."""
        o = np.ascontiguousarray(sensor_origin, np.float64)
        info = capi.KeysInfo()
        capi.check(self._lib.ufomap_map_scan_keys_rgb(self._h, _p(o, C.c_double), d_xyz_ptr, d_rgb_ptr, n, float(max_range), int(depth),
                                                      int(discrete), int(simple_ray_casting), C.byref(info)))
        return info

    def get_keys(self, d_dst_ptr, cap_entries, info):
        capi.check(self._lib.ufomap_map_get_keys(self._h, d_dst_ptr, cap_entries, C.byref(info)))

    def apply_keys(self, d_entries_ptr, info):
        capi.check(self._lib.ufomap_map_apply_keys(self._h, d_entries_ptr, C.byref(info)))

    def apply_keys_batch(self, d_entries_ptrs, infos):
        """Update lists of several depth-0 scans, applied in list order with one walk of the tree."""
        n = len(infos)
        ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(int(p)) for p in d_entries_ptrs])
        arr = (capi.KeysInfo * max(n, 1))()
        for i, k in enumerate(infos):
            arr[i] = k
        capi.check(self._lib.ufomap_map_apply_keys_batch(self._h, ptrs, arr, n))

    def insert_batch(self, comm, origin, d_xyz_ptr, n, max_range=-1.0, depth=0, discrete=True, d_rgb_ptr=None, simple_ray_casting=False,
                     early_stopping=0):
        """This rank's scan of a multi-GPU batch (``ufomap_map_insert_batch_ex``): ray casting here, one RCCL all-gather of the
        scans (bit grids in the steady state, update lists otherwise), all ranks' scans applied in rank order."""
        o = np.ascontiguousarray(origin, dtype=np.float64)
        capi.check(self._lib.ufomap_map_insert_batch_ex(self._h, comm._h, _p(o, C.c_double), C.c_void_p(int(d_xyz_ptr)),
                                                        C.c_void_p(int(d_rgb_ptr)) if d_rgb_ptr else None, n, float(max_range), int(depth), int(discrete),
                                                        int(simple_ray_casting), int(early_stopping)))

    def set_scratch_limit(self, n_bytes):
        """``ufomap_map_set_scratch_limit``: largest dense per-scan grid; scans beyond it take the sparse set of ray cells."""
        capi.check(self._lib.ufomap_map_set_scratch_limit(self._h, int(n_bytes)))

    def insertPointCloudDone(self):
        return bool(capi.check(self._lib.ufomap_map_done(self._h)))

    def insertPointCloudWait(self):
        capi.check(self._lib.ufomap_map_wait(self._h))

    # ---- sensor model setters (occupancy_map_base.h:746-773) -------------------------------------
    def setSensorModel(self, occupied_thres=0.5, free_thres=0.5, prob_hit=0.7, prob_miss=0.4, clamping_thres_min=0.1192,
                       clamping_thres_max=0.971):
        capi.check(self._lib.ufomap_map_set_sensor_model(self._h, occupied_thres, free_thres, prob_hit, prob_miss,
                                                         clamping_thres_min, clamping_thres_max))

    def clear(self):
        capi.check(self._lib.ufomap_map_clear(self._h))

    def reserve(self, n_blocks):
        capi.check(self._lib.ufomap_map_reserve(self._h, n_blocks))

    # ---- read-back in the canonical dump format (same tuple layout as oracle.OracleMap) ----------
    def leaves(self, include_unknown=False):
        n = self._lib.ufomap_map_export_leaves(self._h, int(include_unknown), None, None, None, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        codes, depths, occ, rgb = np.empty(n, np.uint64), np.empty(n, np.uint8), np.empty(n, np.float32), np.zeros((n, 3), np.uint8)
        self._lib.ufomap_map_export_leaves(self._h, int(include_unknown), _p(codes, C.c_uint64), _p(depths, C.c_uint8),
                                           _p(occ, C.c_float), _p(rgb, C.c_uint8), n)
        return codes, depths, occ, rgb

    def inner(self):
        n = self._lib.ufomap_map_export_inner(self._h, None, None, None, None, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        codes, depths, occ = np.empty(n, np.uint64), np.empty(n, np.uint8), np.empty(n, np.float32)
        flags, rgb = np.empty(n, np.uint8), np.zeros((n, 3), np.uint8)
        self._lib.ufomap_map_export_inner(self._h, _p(codes, C.c_uint64), _p(depths, C.c_uint8), _p(occ, C.c_float),
                                          _p(flags, C.c_uint8), _p(rgb, C.c_uint8), n)
        return codes, depths, occ, flags, rgb

    def write(self, filename=None):
        """``Octree::write`` (octree.h:779-868): the reference's .ufo byte stream (uncompressed). Returns the
        bytes; also writes them to ``filename`` when given."""
        n = self._lib.ufomap_map_write(self._h, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        buf = np.empty(n, np.uint8)
        self._lib.ufomap_map_write(self._h, _p(buf, C.c_uint8), n)
        data = buf.tobytes()
        if filename:
            with open(filename, "wb") as f:
                f.write(data)
        return data

    def digest(self, include_unknown=True):
        """Order-independent fingerprint of ``leaves(include_unknown)`` and ``inner()`` (``ufomap_map_digest``):
        (n_leaves, sum, xor, n_inner, sum, xor) as Python ints; ``tests/golden_util.dump_digest`` computes the same
        from a dump."""
        out = np.zeros(6, np.uint64)
        capi.check(self._lib.ufomap_map_digest(self._h, int(include_unknown), _p(out, C.c_uint64)))
        return tuple(int(v) for v in out)

    def minmax_change(self):
        mn, mx = np.empty(3), np.empty(3)
        capi.check(self._lib.ufomap_map_minmax_change(self._h, _p(mn, C.c_double), _p(mx, C.c_double)))
        return mn, mx

    minChange = property(lambda self: self.minmax_change()[0])
    maxChange = property(lambda self: self.minmax_change()[1])

    def resetMinMaxChangeDetection(self):
        capi.check(self._lib.ufomap_map_reset_minmax_change(self._h))

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        capi.check(self._lib.ufomap_map_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        # bytes: the node table as allocated (all slots); bytes_per_block: per LIVE node block (= inner node with its 8 children)
        return dict(inner_nodes=a.value, leaf_nodes=b.value, bytes=c.value, bytes_per_block=(c.value / a.value) if a.value else float("nan"))

    # ---- stage-level outputs / measurement ---------------------------------------------------------
    def _codes(self, fn):
        n = fn(self._h, None, 0)
        if n == C.c_size_t(-1).value:
            capi.check(-2)
        out = np.empty(n, np.uint64)
        fn(self._h, _p(out, C.c_uint64), n)
        return out

    def last_hits(self):
        return self._codes(self._lib.ufomap_map_last_hits)

    def last_misses(self):
        return self._codes(self._lib.ufomap_map_last_misses)

    def last_counts(self):
        c = np.zeros(8, np.uint64)
        capi.check(self._lib.ufomap_map_last_counts(self._h, _p(c, C.c_uint64)))
        keys = ["points", "rays", "steps", "hits", "miss_cells", "blocks_touched", "blocks_created", "oob_dropped"]
        return dict(zip(keys, (int(v) for v in c)))

    def debug(self):
        out = (C.c_uint64 * 64)()
        capi.check(self._lib.ufomap_map_debug(self._h, out, 64))
        return [int(v) for v in out]

    def timeline(self):
        """(records[4096, 8] uint64, newest scan number): ``ufomap_map_timeline`` (option "tstamps" = 1)."""
        out = np.zeros((4096, 8), np.uint64)
        newest = C.c_uint64(0)
        capi.check(self._lib.ufomap_map_timeline(self._h, C.c_void_p(out.ctypes.data), C.c_size_t(out.size), C.byref(newest)))
        return out, int(newest.value)

    def set_option(self, key, value):
        capi.check(self._lib.ufomap_map_set_option(self._h, key.encode(), int(value)))

    def set_profiling(self, on=True):
        capi.check(self._lib.ufomap_map_set_profiling(self._h, int(on)))

    def reset_kernel_times(self):
        capi.check(self._lib.ufomap_map_reset_kernel_times(self._h))

    def kernel_times(self):
        cap = 64
        names = (C.c_char_p * cap)()
        launches = np.zeros(cap, np.uint64)
        ms = np.zeros(cap, np.float64)
        n = capi.check(self._lib.ufomap_map_kernel_times(self._h, names, _p(launches, C.c_uint64), _p(ms, C.c_double), cap))
        return {names[i].decode(): dict(launches=int(launches[i]), total_ms=float(ms[i])) for i in range(min(n, cap))}


class Comm:
    """``ufomap_comm``: the RCCL communicator of the batched multi-GPU path, behind the C ABI (one process per GPU)."""

    ID_BYTES = 128

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int = 0):
        self._lib = capi.load()
        buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._h = self._lib.ufomap_comm_create(buf, world, rank, device)
        if not self._h:
            raise RuntimeError(self._lib.ufomap_last_error().decode())
        self.world, self.rank = world, rank

    @staticmethod
    def unique_id() -> bytes:
        lib = capi.load()
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        capi.check(lib.ufomap_comm_unique_id(buf))
        return bytes(buf)

    def stats(self):
        out = (C.c_uint64 * 4)()
        capi.check(self._lib.ufomap_comm_stats(self._h, out))
        return dict(world=int(out[0]), rank=int(out[1]), slot_bytes=int(out[2]), regrown=int(out[3]))

    def counters(self):
        out = (C.c_uint64 * 4)()
        capi.check(self._lib.ufomap_comm_counters(self._h, out))
        return dict(fast_steps=int(out[0]), repeated_steps=int(out[1]), common_grid=bool(out[2]))

    def close(self):
        if self._h:
            self._lib.ufomap_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OccupancyMap(OccupancyMapBase):
    """``ufo::map::OccupancyMap`` (occupancy_map.h:55)."""
    _color = False


class OccupancyMapColor(OccupancyMapBase):
    """``ufo::map::OccupancyMapColor`` (occupancy_map_color.h:56)."""
    _color = True
