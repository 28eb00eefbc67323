"""toProb(LogitType = float) of the reference is 1 / (1 + std::exp(float)) (occupancy_map_base.h:911): the colour blend
(occupancy_map_color.h:275-277) weighs colours with it. The device evaluates std::exp(float) with ufoExpfRef
(ufomap_amd/csrc/expf_ref.h, the algorithm of glibc's expf); this test compiles those very lines for the host and
compares them with this box's libm -- the function the reference build calls -- for EVERY float32 argument a clamped
log-odds value can produce: all 2.2e9 floats in [-16, 16] (clamping thresholds up to 0.9999999 either way)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))


def _build(shared=False):
    src = os.path.join(HERE, "cpp", "toprob_sweep.c")
    out = os.path.join(HERE, "cpp", "libtoprob_host.so" if shared else "toprob_sweep")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "ufomap_amd", "csrc", "expf_ref.h"))):
        cmd = ["gcc", "-O2", "-ffp-contract=off", src, "-o", out, "-lm"] + (["-shared", "-fPIC", "-DSWEEP_NO_MAIN"] if shared else [])
        subprocess.run(cmd, check=True, capture_output=True)
    return out


def test_expf_of_the_device_equals_libm_expf_for_every_float_in_range():
    exe = _build()
    parts = max(2, min(16, os.cpu_count() or 2))

    def run(k):
        out = subprocess.run([exe, "-16", "16", str(k), str(parts)], check=True, capture_output=True, text=True).stdout.split()
        return int(out[0]), int(out[1]), out[2]

    with ThreadPoolExecutor(parts) as ex:
        res = list(ex.map(run, range(parts)))
    n, bad = sum(r[0] for r in res), sum(r[1] for r in res)
    assert n > 2_100_000_000, n
    assert bad == 0, f"{bad} of {n} arguments differ from libm's expf, e.g. x = {[r[2] for r in res if r[1]][0]}"


def test_expf_special_values():
    import ctypes as C
    import numpy as np
    lib = C.CDLL(_build(shared=True))
    x = np.array([0.0, -0.0, 88.7, 88.73, 100.0, -103.9, -104.0, -200.0, np.inf, -np.inf, 1e-45, -1e-45, 0.84729785, -0.4054651], np.float32)
    a, b = np.empty_like(x), np.empty_like(x)
    for fn, out in ((lib.expf_ref_host, a), (lib.expf_libm, b)):
        fn(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (a, b)
