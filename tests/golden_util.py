"""Load the golden fixtures of tests/golden/ (made by tests/golden/make_golden.py from the reference)."""
import glob
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    """Insert-sequence fixtures (the srv_* files hold the server-loop sequences: see ServerLoop)."""
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if not n.startswith("srv_"))


def server_loop_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "srv_*.npz")))


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


class ServerLoop:
    """A recorded run of the reference through the server's per-message sequence (raw PointCloud2 records + pose
    -> rosToUfo + transform -> insertPointCloudDiscrete -> setValueVolume around the sensor), then point queries.
    `replay(ingest_insert, clear, query, dump)` drives any implementation through it and checks every output."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name, self.z = name, z
        self.meta = json.loads(str(z["meta"]))
        self.params = self.meta["params"]

    def steps(self):
        for i, st in enumerate(self.meta["steps"]):
            yield dict(st, data=self.z[f"d{i}_data"], q=self.z[f"d{i}_q"], t=self.z[f"d{i}_t"], clear_min=self.z[f"d{i}_clear_min"],
                       clear_max=self.z[f"d{i}_clear_max"])

    def check_map(self, m):
        lc, ld, lv, lrgb = m.leaves(True)
        assert len(lc) == self.meta["n_leaves_all"], f"{self.name}: leaf count differs"
        assert digest(lc, ld, lv, lrgb) == self.meta["sha_leaves_all"], f"{self.name}: leaf dump digest differs"
        ic, idp, iv, ifl, irgb = m.inner()
        assert digest(ic, idp, iv, ifl, irgb) == self.meta["sha_inner"], f"{self.name}: inner dump digest differs"
        wb = m.write()
        assert hashlib.sha256(wb).hexdigest() == self.meta["sha_write"], f"{self.name}: byte stream differs from the reference's write()"

    def check_queries(self, query):
        for d in self.meta["query_depths"]:
            lo, st = query(self.z["queries"], d)
            assert np.array_equal(np.asarray(lo).view(np.uint32), self.z[f"q{d}_logodds"].view(np.uint32)), f"{self.name}: query log-odds differ at depth {d}"
            assert np.array_equal(st, self.z[f"q{d}_state"]), f"{self.name}: query states differ at depth {d}"


# ---- order-independent dump fingerprint (the host-side twin of ufomap_map_digest, include/ufomap_hip.h) ----------
_M1, _M2 = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _fold(h):
    if len(h) == 0:
        return 0, 0, 0
    with np.errstate(over="ignore"):
        return len(h), int(np.add.reduce(h, dtype=np.uint64)), int(np.bitwise_xor.reduce(h))


def dump_digest(leaves, inner):
    """(n_leaves, sum, xor, n_inner, sum, xor) of a canonical dump: ``leaves`` = (codes, depths, occ, rgb) and ``inner`` =
    (codes, depths, occ, flags, rgb) as OracleMap / OccupancyMap return them. Equals ``OccupancyMap.digest()``."""
    out = []
    for dump, has_flags in ((leaves, False), (inner, True)):
        codes, depths, occ = dump[0], dump[1], dump[2]
        rgb = dump[4] if has_flags else dump[3]
        key = codes.astype(np.uint64) | (depths.astype(np.uint64) << np.uint64(58))
        val = occ.astype(np.float32).view(np.uint32).astype(np.uint64)
        val |= (rgb[:, 0].astype(np.uint64) | (rgb[:, 1].astype(np.uint64) << np.uint64(8)) | (rgb[:, 2].astype(np.uint64) << np.uint64(16))) << np.uint64(32)
        if has_flags:
            val |= dump[3].astype(np.uint64) << np.uint64(56)
        out.extend(_fold(_mix64(_mix64(key) ^ val)))
    return tuple(out)


def digests():
    """Digest fixtures made from the unmodified reference by tests/golden/make_digests.py (configs too large for a
    full dump in the repo or for the CPU oracle inside a test)."""
    with open(os.path.join(GOLDEN_DIR, "digests.json")) as f:
        return json.load(f)
