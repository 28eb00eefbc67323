"""The C++ host-side mirror (include/ufomap_amd/occupancy_map.hpp): compiles and links against the
C ABI on CPU; on a GPU the example reproduces the known-answer vector of SURVEY.md 8c."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/ufomap_insert_scan"


def _build():
    from ufomap_amd import build
    lib = build.build(force=False, verbose=False)
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "insert_scan.cpp"), lib,
           "-Wl,-rpath," + os.path.dirname(lib), "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True)


SRV = "/tmp/ufomap_server_calls"


def _build_server_calls():
    from ufomap_amd import build
    lib = build.build(force=False, verbose=False)
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "server_calls.cpp"), lib,
           "-Wl,-rpath," + os.path.dirname(lib), "-o", SRV]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_server_calls_compile_against_the_mirror():
    """Every map call of the reference's server.cpp (62-74, 114-120, 150-155, 171-201, 300-391, 468-471), with its
    argument types, compiles against include/ufomap_amd/occupancy_map.hpp; without a GPU the binary fails loudly."""
    from ufomap_amd import capi
    _build_server_calls()
    r = subprocess.run([SRV], capture_output=True, text=True)
    if capi.load().ufomap_device_count() == 0:
        assert r.returncode == 2 and "no CPU fallback" in r.stdout
    else:
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_server_calls_run_on_gpu():
    _build_server_calls()
    r = subprocess.run([SRV], capture_output=True, text=True)
    assert r.returncode == 0 and "server loop ok" in r.stdout, r.stdout + r.stderr


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu():
    from ufomap_amd import capi
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    if capi.load().ufomap_device_count() == 0:
        assert r.returncode == 2 and "no CPU fallback" in r.stdout
    else:
        assert r.returncode == 0


@pytest.mark.gpu
def test_cpp_mirror_kat_on_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, check=True)
    rows = [ln.split() for ln in r.stdout.strip().splitlines()]
    assert [int(x[0]) for x in rows] == [246290604621825, 246290604621832, 246290604621833, 246290604621888, 246290604621889, 246290604621896]
    assert all(x[1] == "0" for x in rows)
    assert [x[2] for x in rows] == ["-0.405465"] * 5 + ["0.441833"]
