"""Developer aid: the mixed-path sequence of tests/test_gpu_vol.py::test_volume_path_walk_in_flight_and_other_paths, prefix by prefix."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_vol import _maps, _wander, _insert, _kind
seq = _wander(9, spread=0.5)
for n in range(2, 10):
    g, o = _maps(kind="port", resolution=0.16)
    for i, (origin, xyz) in enumerate(seq[:n]):
        forced = i in (0, 1, 4, 5, 8)
        g.set_option("vol", 2 if forced else 1)
        g.set_option("spec", 0 if forced or i == 6 else 1)
        _insert(g, origin, xyz, 12.0, True, async_=True)
        o.insert(origin, xyz, max_range=12.0, discrete=True)
    g.insertPointCloudWait()
    gl, ol = g.leaves(True), o.leaves(True)
    d = g.debug()
    print("prefix", n, "leaves", len(gl[0]), len(ol[0]), "same", len(gl[0]) == len(ol[0]) and all(np.array_equal(a, b) for a, b in zip(gl, ol)), "vol", d[48:51], "fast", d[61], flush=True)
