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
