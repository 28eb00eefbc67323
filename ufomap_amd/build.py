"""Build the HIP library in-tree:  python -m ufomap_amd.build  (or __graft_entry__.build()).

One translation unit, gfx950 only.  -ffp-contract=off is part of the arithmetic contract
(SURVEY.md 8a'): hipcc contracts a*b+c to FMA by default, which would change DDA keys.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libufomap_hip.so")
SOURCES = ["ufomap_hip.hip"]
HEADERS = ["geom.h", "table.h", "expf_ref.h", "scan_kernels.h", "map_kernels.h", "fast_kernels.h", "vol_kernels.h", "host_fast_path.inl", "host_vol.inl", "host_multi_gpu.inl",
           "host_serialise.inl", os.path.join("..", "..", "include", "ufomap_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


EXAMPLE_SRC = os.path.join(os.path.dirname(HERE), "examples", "bench_loop.cpp")
EXAMPLE_BIN = os.path.join(os.path.dirname(HERE), "examples", "bench_loop")


def build_bench_loop(verbose: bool = True) -> str:
    """examples/bench_loop: the bench's headline loop in C++ through the C ABI (bench.py leg "host_cxx")."""
    build(verbose=verbose)
    deps = [EXAMPLE_SRC, LIB, os.path.join(os.path.dirname(HERE), "include", "ufomap_hip.h")]
    if os.path.exists(EXAMPLE_BIN) and all(os.path.getmtime(d) <= os.path.getmtime(EXAMPLE_BIN) for d in deps):
        return EXAMPLE_BIN
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-I" + os.path.join(os.path.dirname(HERE), "include"), EXAMPLE_SRC, "-L" + CSRC, "-lufomap_hip", "-Wl,-rpath," + CSRC, "-o", EXAMPLE_BIN]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return EXAMPLE_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
