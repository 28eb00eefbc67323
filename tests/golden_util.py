"""Load the golden fixtures of tests/golden/ (made by tests/golden/make_golden.py from the reference)."""
import glob
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.params = self.meta["params"]
        self.z = z

    def scans(self):
        """Yield (origin, xyz, rgb|None, kwargs) in integration order."""
        for sc in self.meta["scans"]:
            k = sc["data"]
            rgb = self.z[k + "_rgb"] if sc["has_rgb"] else None
            yield self.z[k + "_origin"], self.z[k + "_xyz"], rgb, sc["kwargs"]

    def check(self, m):
        """Assert that map object `m` (OracleMap-like: leaves/inner/minmax_change) equals the reference dump."""
        kc, kd, kv, krgb = m.leaves(False)
        assert np.array_equal(kc, self.z["leaf_codes"]), f"{self.name}: known-leaf codes differ"
        assert np.array_equal(kd, self.z["leaf_depths"]), f"{self.name}: known-leaf depths differ"
        assert np.array_equal(kv, self.z["leaf_occ"]), f"{self.name}: known-leaf log-odds differ (bitwise)"
        assert np.array_equal(krgb, self.z["leaf_rgb"]), f"{self.name}: known-leaf colours differ"
        lc, ld, lv, lrgb = m.leaves(True)
        assert len(lc) == self.meta["n_leaves_all"], f"{self.name}: leaf count (incl. unknown) differs"
        assert digest(lc, ld, lv, lrgb) == self.meta["sha_leaves_all"], f"{self.name}: full leaf dump digest differs"
        ic, idp, iv, ifl, irgb = m.inner()
        assert len(ic) == self.meta["n_inner"], f"{self.name}: inner-node count differs"
        assert digest(ic, idp, iv, ifl, irgb) == self.meta["sha_inner"], f"{self.name}: inner dump digest differs"
        if "sha_write" in self.meta and hasattr(m, "write"):
            wb = m.write()
            assert len(wb) == self.meta["write_size"], f"{self.name}: byte-stream size differs from the reference's write()"
            assert hashlib.sha256(wb).hexdigest() == self.meta["sha_write"], f"{self.name}: byte stream differs from the reference's write()"
        mn, mx = m.minmax_change()
        assert np.array_equal(mn, self.z["min_change"]) and np.array_equal(mx, self.z["max_change"]), f"{self.name}: change AABB differs"
