"""ctypes binding of include/ufomap_hip.h (the C-ABI of the HIP library).

The library is built in-tree by ufomap_amd/build.py (csrc/libufomap_hip.so) and loaded from
there -- never from site-packages -- so that the driver sees which native code ran.  Loading
fails loudly when the .so is missing; there is no Python or CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UFOMAP_HIP_LIB") or os.path.join(HERE, "csrc", "libufomap_hip.so")  # (the variable: developer A/B of two builds)

UFOMAP_OK = 0
ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED, ERR_RUNAWAY, ERR_CAPACITY = -1, -2, -3, -4, -5

# every symbol include/ufomap_hip.h declares (tests/test_capi_symbols.py checks the .so against this
# list AND the list against the header)
SYMBOLS = [
    "ufomap_last_error", "ufomap_device_count", "ufomap_version", "ufomap_map_create",
    "ufomap_map_destroy", "ufomap_map_clear", "ufomap_map_reserve", "ufomap_map_set_scratch_limit",
    "ufomap_map_set_sensor_model", "ufomap_map_insert", "ufomap_map_insert_device", "ufomap_map_insert_pointcloud2", "ufomap_map_set_value_volume", "ufomap_map_query", "ufomap_map_clamping_thres",
    "ufomap_map_wait", "ufomap_map_done", "ufomap_map_export_leaves", "ufomap_map_export_inner",
    "ufomap_map_write", "ufomap_map_digest", "ufomap_map_minmax_change", "ufomap_map_reset_minmax_change", "ufomap_map_stats",
    "ufomap_map_last_hits", "ufomap_map_last_misses", "ufomap_map_last_counts",
    "ufomap_map_set_profiling", "ufomap_map_kernel_times", "ufomap_map_reset_kernel_times",
    "ufomap_map_clear_to", "ufomap_map_get_sensor_model", "ufomap_map_set_model_value", "ufomap_map_set_occupied_free_thres",
    "ufomap_map_set_value_volume_ch", "ufomap_map_enable_change_detection", "ufomap_map_reset_change_detection", "ufomap_map_changes",
    "ufomap_map_enable_minmax_change_detection", "ufomap_map_iterate", "ufomap_map_write_ex", "ufomap_map_read", "ufomap_map_read_data",
    "ufomap_map_scan_keys", "ufomap_map_scan_keys_rgb", "ufomap_map_get_keys", "ufomap_map_apply_keys", "ufomap_map_apply_keys_batch",
    "ufomap_comm_unique_id", "ufomap_comm_create", "ufomap_comm_from_nccl", "ufomap_comm_destroy", "ufomap_comm_stats", "ufomap_comm_counters", "ufomap_map_insert_batch", "ufomap_map_insert_batch_ex", "ufomap_dev_expf", "ufomap_map_timeline",
    "ufomap_map_stream", "ufomap_map_debug", "ufomap_map_set_option", "ufomap_alloc_counters",
]

_lib = None


class KeysInfo(C.Structure):
    """``ufomap_keys_info`` of include/ufomap_hip.h (header of one scan's update list)."""
    _fields_ = [("n_hit", C.c_uint32), ("n_miss", C.c_uint32), ("nb_hit", C.c_int32 * 3), ("nb_miss", C.c_int32 * 3),
                ("depth", C.c_uint32), ("reserved", C.c_uint32)]

    WORDS = 10  # as a flat int32 vector for exchange between ranks

    @property
    def list_bytes(self):
        """Size of the list on the wire: 16-byte records + (colour maps) 32 bytes of colours per hit record."""
        return (self.n_hit + self.n_miss) * 16 + (self.n_hit * 32 if (self.reserved & 2) else 0)

    def to_list(self):
        return [self.n_hit, self.n_miss, *self.nb_hit, *self.nb_miss, self.depth, self.reserved]

    @classmethod
    def from_list(cls, v):
        k = cls()
        k.n_hit, k.n_miss = int(v[0]), int(v[1])
        for a in range(3):
            k.nb_hit[a] = int(v[2 + a])
            k.nb_miss[a] = int(v[5 + a])
        k.depth, k.reserved = int(v[8]), int(v[9])
        return k


class UfomapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ufomap_hip error {code}: {msg}")
        self.code = code


def load():
    """Load csrc/libufomap_hip.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -m ufomap_amd.build` (no CPU fallback exists)")
    lib = C.CDLL(LIB_PATH)
    vp, sz, dbl, u64p, u8p, f32p, f64p = C.c_void_p, C.c_size_t, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.ufomap_last_error.restype = C.c_char_p
    lib.ufomap_version.restype = C.c_char_p
    lib.ufomap_device_count.restype = C.c_int
    lib.ufomap_map_create.restype = vp
    lib.ufomap_map_create.argtypes = [dbl, C.c_uint, C.c_int] + [dbl] * 6 + [C.c_int, C.c_int]
    lib.ufomap_map_destroy.argtypes = [vp]
    lib.ufomap_map_destroy.restype = None
    lib.ufomap_map_clear.argtypes = [vp]
    lib.ufomap_map_reserve.argtypes = [vp, sz]
    lib.ufomap_map_set_scratch_limit.argtypes = [vp, sz]
    lib.ufomap_map_set_sensor_model.argtypes = [vp] + [dbl] * 6
    ins = [vp, f64p, vp, vp, sz, dbl, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int]
    lib.ufomap_map_insert.argtypes = ins
    lib.ufomap_map_insert_device.argtypes = ins
    lib.ufomap_map_set_value_volume.argtypes = [vp, f64p, f64p, dbl, C.c_uint]
    lib.ufomap_map_query.argtypes = [vp, vp, C.c_int, sz, C.c_uint, f32p, u8p]
    lib.ufomap_map_clamping_thres.argtypes = [vp, f64p, f64p]
    lib.ufomap_map_wait.argtypes = [vp]
    lib.ufomap_map_done.argtypes = [vp]
    lib.ufomap_map_export_leaves.restype = sz
    lib.ufomap_map_export_leaves.argtypes = [vp, C.c_int, u64p, u8p, f32p, u8p, sz]
    lib.ufomap_map_export_inner.restype = sz
    lib.ufomap_map_export_inner.argtypes = [vp, u64p, u8p, f32p, u8p, u8p, sz]
    lib.ufomap_map_digest.argtypes = [vp, C.c_int, u64p]
    lib.ufomap_map_write.restype = sz
    lib.ufomap_map_write.argtypes = [vp, u8p, sz]
    lib.ufomap_map_minmax_change.argtypes = [vp, f64p, f64p]
    lib.ufomap_map_reset_minmax_change.argtypes = [vp]
    lib.ufomap_map_stats.argtypes = [vp, u64p, u64p, u64p]
    lib.ufomap_map_last_hits.restype = sz
    lib.ufomap_map_last_hits.argtypes = [vp, u64p, sz]
    lib.ufomap_map_last_misses.restype = sz
    lib.ufomap_map_last_misses.argtypes = [vp, u64p, sz]
    lib.ufomap_map_last_counts.argtypes = [vp, u64p]
    lib.ufomap_map_set_profiling.argtypes = [vp, C.c_int]
    lib.ufomap_map_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), u64p, f64p, C.c_int]
    lib.ufomap_map_reset_kernel_times.argtypes = [vp]
    lib.ufomap_map_scan_keys.argtypes = [vp, f64p, vp, sz, dbl, C.c_uint, C.c_int, C.c_int, C.POINTER(KeysInfo)]
    lib.ufomap_map_scan_keys_rgb.argtypes = [vp, f64p, vp, vp, sz, dbl, C.c_uint, C.c_int, C.c_int, C.POINTER(KeysInfo)]
    lib.ufomap_map_get_keys.argtypes = [vp, vp, sz, C.POINTER(KeysInfo)]
    lib.ufomap_map_insert_pointcloud2.argtypes = [vp, f64p, f64p, vp, C.c_int, sz, C.c_uint32] + [C.c_int] * 6 + [
        dbl, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_int]
    lib.ufomap_map_apply_keys.argtypes = [vp, vp, C.POINTER(KeysInfo)]
    lib.ufomap_map_apply_keys_batch.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(KeysInfo), C.c_int]
    lib.ufomap_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    lib.ufomap_comm_create.restype = vp
    lib.ufomap_comm_create.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int]
    lib.ufomap_comm_from_nccl.restype = vp
    lib.ufomap_comm_from_nccl.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.ufomap_comm_destroy.restype = None
    lib.ufomap_comm_destroy.argtypes = [vp]
    lib.ufomap_comm_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ufomap_comm_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.ufomap_map_insert_batch.argtypes = [vp, vp, f64p, vp, vp, sz, dbl, C.c_uint, C.c_int]
    lib.ufomap_map_insert_batch_ex.argtypes = [vp, vp, f64p, vp, vp, sz, dbl, C.c_uint, C.c_int, C.c_int, C.c_uint]
    lib.ufomap_map_timeline.argtypes = [vp, vp, sz, u64p]
    lib.ufomap_map_clear_to.argtypes = [vp, dbl, C.c_uint]
    lib.ufomap_map_get_sensor_model.argtypes = [vp, f64p]
    lib.ufomap_map_set_model_value.argtypes = [vp, C.c_int, dbl]
    lib.ufomap_map_set_occupied_free_thres.argtypes = [vp, dbl, dbl]
    lib.ufomap_map_set_value_volume_ch.argtypes = [vp, f64p, f64p, dbl, C.c_uint]
    lib.ufomap_map_enable_change_detection.argtypes = [vp, C.c_int]
    lib.ufomap_map_reset_change_detection.argtypes = [vp]
    lib.ufomap_map_changes.restype = sz
    lib.ufomap_map_changes.argtypes = [vp, u64p, u8p, sz]
    lib.ufomap_map_enable_minmax_change_detection.argtypes = [vp, C.c_int]
    lib.ufomap_map_iterate.restype = sz
    lib.ufomap_map_iterate.argtypes = [vp, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, u64p, u8p, f32p, u8p, u8p, sz]
    lib.ufomap_map_write_ex.restype = sz
    lib.ufomap_map_write_ex.argtypes = [vp, f64p, f64p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, u8p, sz, C.POINTER(C.c_longlong)]
    lib.ufomap_map_read.argtypes = [vp, u8p, sz, f64p, C.POINTER(C.c_uint)]
    lib.ufomap_map_read_data.argtypes = [vp, u8p, sz, f64p, f64p, dbl, C.c_uint, C.c_int, C.c_int]
    lib.ufomap_map_debug.argtypes = [vp, u64p, C.c_int]
    lib.ufomap_map_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    lib.ufomap_map_stream.restype = vp
    lib.ufomap_map_stream.argtypes = [vp]
    lib.ufomap_alloc_counters.restype = None
    lib.ufomap_alloc_counters.argtypes = [u64p]
    _lib = lib
    return lib


def alloc_counters():
    """``ufomap_alloc_counters``: device allocations of this process so far."""
    out = (C.c_uint64 * 5)()
    load().ufomap_alloc_counters(out)
    return dict(zip(("mallocs", "frees", "bytes", "host_ns", "rehashes"), (int(v) for v in out)))


def check(rc):
    if rc < 0:
        raise UfomapError(rc, load().ufomap_last_error().decode())
    return rc
