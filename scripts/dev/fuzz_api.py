"""Developer aid (GPU box): random sequences of calls on ONE map -- host / device / PointCloud2 clouds, synchronous and asynchronous, insert
depths, continuous and discrete, robot clearing, batch steps over a one-rank communicator that comes and goes, byte streams, clear -- against
the CPU checker after every few calls (round 6: a hand-over set that carried state from one kind of call into the next was what crashed the
bench; this looks for the next one).   python scripts/dev/fuzz_api.py [seeds=6] [ops=120] [first_seed=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WORLD = int(os.environ.get("FUZZ_WORLD", "1"))  # > 1: this process plays rank 0 of WORLD (tests/cpp/rccl_shim.cpp: every slot of an all-gather is a copy of the caller's)
if WORLD > 1:
    import subprocess
    shim_src, shim = os.path.join(ROOT, "tests", "cpp", "rccl_shim.cpp"), os.path.join(ROOT, "tests", "cpp", "librccl_shim.so")
    if not os.path.exists(shim) or os.path.getmtime(shim) < os.path.getmtime(shim_src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", shim_src, "-o", shim, "-lrt"], check=True)
    os.environ["UFOMAP_RCCL_LIB"] = shim
    os.environ["UFOMAP_SHIM_REPLICATE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import OracleMap, RunawayRay  # noqa: E402
from oracle import ingest as oracle_ingest  # noqa: E402
from ufomap_amd import OccupancyMap, OccupancyMapColor, PointCloud, PointCloudColor, scans  # noqa: E402
from ufomap_amd.occupancy_map import Comm  # noqa: E402


def same_dump(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 120
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
check_from = int(os.environ.get("FUZZ_CHECK_FROM", "1000000"))  # (from this call on, the maps are compared after every FUZZ_CHECK_EVERY-th call)
check_every = int(os.environ.get("FUZZ_CHECK_EVERY", "1"))
print_all = bool(os.environ.get("FUZZ_PRINT"))
ext_ops = bool(os.environ.get("FUZZ_EXT")) or bool(os.environ.get("FUZZ_COLOR"))  # more kinds of calls: early stopping / fixed-step casting, point queries
chg = os.environ.get("FUZZ_CHG", "")  # "codes": the per-code change set, "box": the min / max change box -- compared (and reset) with every comparison of the maps
RES = float(os.environ.get("FUZZ_RES", "0.16"))  # leaf size; FUZZ_RANGE: max_range (a fine map with a long range takes the ray grid beyond LDS)
RANGE = float(os.environ.get("FUZZ_RANGE", "12.0"))
color = bool(os.environ.get("FUZZ_COLOR"))  # OccupancyMapColor with coloured clouds (discrete only: the reference's continuous form does not compile with colours)
skip = set(filter(None, os.environ.get("FUZZ_SKIP", "").split(",")))  # kinds of calls that are left out (both maps)
kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libufo_ref.so")) else "port"


def check(g, o, what):
    g.insertPointCloudWait()
    ok = same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner())
    if not ok:
        raise AssertionError(what)
    if chg == "codes":
        ga, oa = g.changes(), o.changes()  # (the reference's change set is an unordered set: compared as sets)
        sg = set(zip(ga[1].tolist(), ga[0].tolist()))
        so = set(zip(oa[1].tolist(), oa[0].tolist()))
        # (... of nodes: after scans at insert depth > 0 the reference's set holds several Codes whose low bits differ for ONE node -- the
        # checker's export lists them all, 48 452 entries for 43 974 nodes in one sequence; the library reports each node once)
        if sg != so:
            only_g, only_o = sorted(sg - so), sorted(so - sg)
            by_d = lambda xs: {d: sum(1 for x in xs if x[0] == d) for d in sorted({x[0] for x in xs})}  # noqa: E731
            raise AssertionError(what + f" (the change sets: {len(sg)} vs {len(so)} codes, arrays of {len(ga[0])} vs {len(oa[0])}; only here by depth {by_d(only_g)}, only in the checker {by_d(only_o)}; "
                                        f"e.g. {only_g[:3]} / {only_o[:3]})")
        g.resetChangeDetection()
        o.resetChangeDetection()
    elif chg == "box":
        a, b = g.minmax_change(), o.minmax_change()
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            raise AssertionError(what + f" (the change boxes: {a} vs {b})")
        # (the boxes accumulate: the checker has no reset)


bad = 0
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(seed)
    g, o = (OccupancyMapColor if color else OccupancyMap)(RES), OracleMap(RES, kind=kind, color=color)
    if chg == "codes":
        g.enableChangeDetection(True)
        o.enableChangeDetection(True)
    elif chg == "box":
        g.enableMinMaxChangeDetection(True)
        o.enableMinMaxChangeDetection(True)
    for kv in filter(None, os.environ.get("FUZZ_OPTS", "").split(",")):  # options for the map: "early_map=0,lazy_done=0"
        g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    comm, keep, log = None, [], []
    pose = np.array(scans.lidar_pose(int(rng.integers(0, 4))), dtype=np.float64)
    try:
        for step in range(n_ops):
            if ext_ops:
                op = rng.choice(["host", "host", "dev", "dev", "pc2", "batch", "batch", "vol", "bytes", "wait", "clear", "comm", "big", "cont", "depth", "es", "query", "read"],
                                p=[0.12, 0.1, 0.11, 0.09, 0.08, 0.1, 0.07, 0.07, 0.03, 0.04, 0.01, 0.04, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02])
            else:  # (the table the seeds of tests/test_gpu_batch.py: test_random_call_sequences_against_the_checker were found with)
                op = rng.choice(["host", "host", "dev", "dev", "pc2", "batch", "batch", "vol", "bytes", "wait", "clear", "comm", "big", "cont", "depth"],
                                p=[0.14, 0.1, 0.12, 0.1, 0.08, 0.1, 0.08, 0.07, 0.04, 0.05, 0.01, 0.04, 0.03, 0.02, 0.02])
            if color and op in ("cont", "pc2"):
                op = "host"
            if chg and op == "clear":  # (a cleared map is a NEW checker map here, whose change set / box starts empty: the reference's clear() keeps them)
                op = "wait"
            if rng.random() < 0.12:
                pose = pose + rng.normal(0, 1.5, 3) * np.array([1, 1, 0.1])  # (a jump: the predicted grid misses, the scan is repeated)
            else:
                pose = pose + rng.normal(0, 0.04, 3) * np.array([1, 1, 0.2])
            origin = tuple(pose)
            asyn = bool(rng.random() < 0.7)
            r2 = np.random.default_rng([seed, step])  # (the call's own parameters: skipping a kind of call leaves the rest of the sequence as it is)
            beams, az = (64, 2048) if op == "big" else (16, int(r2.choice([128, 512])))
            _, xyz, rgb = scans.lidar64(beams=beams, azimuths=az, origin=origin, seed=int(r2.integers(1 << 30)), colored=color)
            if r2.random() < 0.05:
                k_pts = int(r2.integers(0, 3))
                xyz, rgb = xyz[:k_pts], (rgb[:k_pts] if color else None)
            if not color:
                rgb = None

            def cloud(a, c):
                return PointCloudColor(a, c) if color else PointCloud(a)
            log.append((step, op, asyn, len(xyz)))
            if op in skip:
                continue
            if op in ("host", "big"):
                buf, cbuf = xyz.copy(), (rgb.copy() if color else None)
                g.insertPointCloudDiscrete(origin, cloud(buf, cbuf), RANGE, 0, False, 0, asyn)
                buf[:] = 7.0
                if color:
                    cbuf[:] = 3
                o.insert(origin, xyz, rgb, max_range=RANGE, discrete=True)
            elif op == "cont":
                g.insertPointCloud(origin, PointCloud(xyz), RANGE, 0, False, 0, asyn)
                o.insert(origin, xyz, max_range=RANGE, discrete=False)
            elif op == "depth":
                d = int(r2.integers(1, 3))
                g.insertPointCloudDiscrete(origin, cloud(xyz, rgb), RANGE, d, False, 0, asyn)
                o.insert(origin, xyz, rgb, max_range=RANGE, discrete=True, depth=d)
            elif op == "es":
                simple, es = bool(r2.random() < 0.5), int(r2.integers(0, 4))
                log[-1] = log[-1] + (f"simple={simple} es={es}",)
                g.insertPointCloudDiscrete(origin, cloud(xyz, rgb), RANGE, 0, simple, es, asyn)
                o.insert(origin, xyz, rgb, max_range=RANGE, discrete=True, simple_ray_casting=simple, early_stopping=es)
            elif op == "query":
                q = np.concatenate([xyz[::7] + r2.normal(0, 0.1, xyz[::7].shape), r2.uniform(-15, 15, (500, 3))]) if len(xyz) else r2.uniform(-15, 15, (500, 3))
                qd = int(r2.integers(0, 4))
                a, b = g.query(q, qd), o.query(q, qd)
                if not (np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])):
                    raise AssertionError(f"point queries at depth {qd} differ")
            elif op == "dev":
                d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
                dc = torch.from_numpy(np.ascontiguousarray(rgb)).cuda() if color else None
                keep.extend([d, dc])
                g.insert_device(origin, d.data_ptr() if len(xyz) else 0, (dc.data_ptr() if len(xyz) else 0) if color else None, len(xyz), RANGE, 0, discrete=True, async_=asyn)
                o.insert(origin, xyz, rgb, max_range=RANGE, discrete=True)
            elif op == "pc2":
                rec = np.zeros((len(xyz), 4), np.float32)
                rec[:, :3] = (xyz - np.asarray(origin)[None, :]).astype(np.float32)
                raw = np.ascontiguousarray(rec).view(np.uint8).reshape(-1)
                if len(xyz):
                    g.insertPointCloud2(np.asarray(origin), np.array([1.0, 0, 0, 0]), raw, 16, (0, 4, 8), None, RANGE, 0, True, False, 0, asyn)
                    xyz32, _ = oracle_ingest(raw, 16, (0, 4, 8), None, np.array([1.0, 0, 0, 0]), np.asarray(origin), kind)  # (the reference's rosToUfo + transform)
                    o.insert(origin, xyz32, max_range=RANGE, discrete=True)
            elif op == "batch":
                if comm is None:
                    comm = Comm(Comm.unique_id(), WORLD, 0, 0)
                g.set_option("async_apply", int(asyn))
                for _ in range(int(r2.integers(1, 5))):
                    d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
                    dc = torch.from_numpy(np.ascontiguousarray(rgb)).cuda() if color else None
                    keep.extend([d, dc])
                    dep = int(r2.integers(0, 3)) if (r2.random() < 0.1 and not color) else 0
                    g.insert_batch(comm, origin, d.data_ptr() if len(xyz) else 0, len(xyz), RANGE, dep, True, (dc.data_ptr() if len(xyz) else 0) if color else None)
                    for _w in range(WORLD):  # (the step applies the scans of ranks 0 .. WORLD-1 in this order: here the same scan WORLD times)
                        o.insert(origin, xyz, rgb, max_range=RANGE, discrete=True, depth=dep)
            elif op == "comm":
                if comm is not None:
                    g.insertPointCloudWait()
                    comm.close()
                    comm = None
            elif op == "vol":
                ext = r2.uniform(0.2, 1.8, 3)
                md = int(r2.integers(0, 3))
                val = float(r2.choice([g.getClampingThresMin(), 0.5, 0.3]))
                g.setValueVolume(pose - ext, pose + ext, val, md)
                o.setValueVolume(pose - ext, pose + ext, val, md)
            elif op == "bytes":
                g.insertPointCloudWait()
                a, b = g.write(), o.write()
                if a != b:
                    same = same_dump(g.leaves(True), o.leaves(True)) and same_dump(g.inner(), o.inner())
                    a2 = g.write()
                    na, nb = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
                    k = min(len(na), len(nb))
                    d = np.nonzero(na[:k] != nb[:k])[0]
                    raise AssertionError(f"byte streams differ: {len(a)} vs {len(b)} bytes, first difference at {int(d[0]) if len(d) else k} of {len(d)}; maps equal: {same}; "
                                         f"a second write equals the checker's: {a2 == b}; debug {g.debug()[40:64]}")
                ext = r2.uniform(0.5, 4.0, 3)
                a = g.write_ex(aabb=(pose - ext, pose + ext), compress=False, min_depth=int(r2.integers(0, 2)), header=False)[0]
                log[-1] = log[-1] + (len(a),)
            elif op == "read":
                # a message of the map's own (a sub-volume, as the server publishes it) merged back after the next update has changed things
                box = (pose - r2.uniform(0.5, 3.0, 3), pose + r2.uniform(0.5, 3.0, 3))
                blob, usz = o.write_ex(aabb=box, compress=bool(r2.random() < 0.3), min_depth=0, header=False)
                comp = usz != len(blob) and usz > 0
                g.readData(blob, RES, 16, usz if usz > 0 else len(blob), comp, box)
                o.readData(blob, RES, 16, usz if usz > 0 else len(blob), comp, box)
            elif op == "wait":
                check(g, o, f"seed {seed} step {step}: maps differ")
            elif op == "clear":
                g.insertPointCloudWait()
                g.clear()
                o = OracleMap(RES, kind=kind, color=color)
            if len(keep) > 64:
                g.insertPointCloudWait()
                keep.clear()
            if step >= check_from and (step - check_from) % check_every == 0:
                check(g, o, f"seed {seed} step {step} ({op}): maps differ")
        check(g, o, f"seed {seed}: maps differ at the end")
        if g.write() != o.write():
            raise AssertionError("byte streams differ at the end")
        print(f"seed {seed}: ok ({n_ops} calls, {g.debug()[61]} steady-state scans)", flush=True)
    except RunawayRay:
        print(f"seed {seed}: skipped (a runaway ray in the checker)", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"seed {seed}: FAILED at {log[-1] if log else None}: {e!r}\n  last calls: {log if print_all else log[-12:]}", flush=True)
    finally:
        if comm is not None:
            try:
                g.insertPointCloudWait()
            except Exception:  # noqa: BLE001
                pass
            comm.close()
        del g, o
print(f"{bad} of {n_seeds} sequences failed")
sys.exit(1 if bad else 0)
