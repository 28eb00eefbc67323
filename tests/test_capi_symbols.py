"""CPU checks of the drop-in boundary: the in-tree C-ABI library loads, exports every symbol that
include/ufomap_hip.h declares, and fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ufomap_amd import build, capi
    build.build(force=False, verbose=False)
    return capi.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ufomap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ufomap_[a-z_0-9]+)\s*\(", text)))


def test_library_is_in_tree_and_loads(lib):
    from ufomap_amd import capi
    assert capi.LIB_PATH.startswith(ROOT) and os.path.exists(capi.LIB_PATH)
    assert lib.ufomap_version().decode().startswith("ufomap_amd")


def test_every_declared_symbol_is_exported(lib):
    from ufomap_amd import capi
    declared = _declared_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/ufomap_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == declared, "ufomap_amd/capi.py SYMBOLS out of sync with the header"


def test_keys_info_layout_matches_header(lib):
    from ufomap_amd import capi
    assert ctypes.sizeof(capi.KeysInfo) == 4 * capi.KeysInfo.WORDS == 40
    k = capi.KeysInfo.from_list(list(range(10)))
    assert k.to_list() == list(range(10))


def test_no_cpu_fallback_without_device(lib):
    """Without a GPU the constructor must fail with a device error -- never silently run on the CPU."""
    if lib.ufomap_device_count() > 0:
        pytest.skip("a HIP device is present")
    from ufomap_amd import OccupancyMap, capi
    with pytest.raises(capi.UfomapError) as e:
        OccupancyMap(0.16)
    assert e.value.code == capi.ERR_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_constructor_argument_errors_come_first(lib):
    """depth_levels outside [2,21] is an invalid_argument in the reference (octree.h:931-935)."""
    from ufomap_amd import OccupancyMap
    with pytest.raises(ValueError):
        OccupancyMap(0.1, depth_levels=1)
    with pytest.raises(ValueError):
        OccupancyMap(0.1, depth_levels=22)


def test_product_code_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "ufomap_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp")):
                src = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "ufo_oracle" in src:
                    offenders.append(os.path.join(base, f))
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            if "ufo_oracle" in open(os.path.join(base, f), errors="ignore").read():
                offenders.append(os.path.join(base, f))
    assert offenders == []


def test_header_is_plain_c(tmp_path):
    """include/ufomap_hip.h is the drop-in boundary: it must compile as C99 (plain pointers and sizes, no C++), and a C
    host of the multi-GPU batch entry must compile and link against the library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.c"
    src.write_text(
        '#include <stdio.h>\n#include "ufomap_hip.h"\n'
        "int main(void) {\n"
        "  uint8_t id[UFOMAP_COMM_ID_BYTES];\n"
        "  ufomap_keys_info k; k.n_hit = 0;\n"
        "  if (ufomap_device_count() <= 0) { printf(\"no device: %s\\n\", ufomap_last_error()); return 0; }\n"
        "  if (ufomap_comm_unique_id(id)) return 2;\n"
        "  ufomap_comm* c = ufomap_comm_create(id, 1, 0, 0);\n"
        "  ufomap_map* m = ufomap_map_create(0.16, 16, 1, 0.5, 0.5, 0.7, 0.4, 0.1192, 0.971, 0, 0);\n"
        "  double o[3] = {0, 0, 0};\n"
        "  int rc = (c && m) ? ufomap_map_insert_batch(m, c, o, 0, 0, 0, 20.0, 0, 1) : 3;\n"
        "  ufomap_map_destroy(m); ufomap_comm_destroy(c); (void)k; return rc;\n}\n")
    exe = tmp_path / "host"
    lib_dir = os.path.join(root, "ufomap_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-l:libufomap_hip.so", "-Wl,-rpath," + lib_dir], check=True)
    # (without a GPU the program reports that and exits 0; with one it runs a world-1 batch step)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cxx_bench_loop_builds_and_fails_loudly_without_a_device(lib):
    """examples/bench_loop.cpp (bench.py's leg "host_cxx": the headline loop driven from C++ through the C ABI) links against the
    in-tree library; without a HIP device it ends with an error code instead of pretending to measure anything."""
    import subprocess
    from ufomap_amd import build
    exe = build.build_bench_loop(verbose=False)
    assert os.path.exists(exe)
    if lib.ufomap_device_count() > 0:
        pytest.skip("a HIP device is present: the bench runs the loop")
    r = subprocess.run([exe, "/nonexistent", "1", "1", "0.01", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
