#!/usr/bin/env python
"""bench.py -- headline benchmark of the scan-integration hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (N > 1, one rank per GPU)

One *step* = one pass of the hot path over one scan: BASELINE.json configs[1] -- a synthetic 64-beam LiDAR scan
(131 072 points), 16 cm leaf, 20 m max range, insertPointCloudDiscrete (discrete integrator + free-space ray cast)
into a GPU-resident linear-hashed octree.  The sensor MOVES: step i integrates the scan taken at pose i mod 8 of
BASELINE configs[3] (3 m apart, seeds 100 + pose), starting from a FRESH map in every repetition, so that node blocks
are created, values change and summaries propagate -- what a mapping server sees, not a static sensor re-integrating
one scan into a saturated map (kept as the extra key `resident_static`).  The ray grid the scans are enqueued on is
predicted from the scans before; after the first pass over the 8 poses it is their common hull and survives the
map's clear(), so in the timed region no scan is repeated for a misprediction (`pipeline.predicted_grid_repeats`).

Timed region = W warm-up steps into a cleared map, barrier + synchronise, EXACTLY K steps, synchronise + barrier.
With the driver's K = 20 that is only a few milliseconds, so the region is REPEATED (map cleared, same W + K
steps) until at least 0.5 s of timed steps have accumulated; `value` is total points / total timed seconds over
all repetitions (`repeats`, `timed_region_s`; `ms_per_step_median_rep` for the spread). Python's cyclic garbage
collector is off while the legs run (as in `timeit`): with torch imported one full collection is a 35 ms pause of the
calling thread, which a C++ caller of the library does not have.

Which number is which.  `value` is measured with the clouds ALREADY RESIDENT IN HBM when the timed region starts
(ufomap_map_insert_device, async=true), as the task's measurement rules prescribe.  SURVEY.md 8(d) defines the metric
for the call the reference's server makes -- host cloud in, H2D inside the timed call: that figure is
`value_incl_h2d` (= the `host_pointer` leg); `pointcloud2` is the same with the raw float32 records of a
sensor_msgs/PointCloud2 (16 B per point over PCIe, conversion and transform fused into the first kernel), and
`server_loop` the server's whole per-message sequence (ingest + integrate, robot clearing, serialising the changed
part of the map; server.cpp:114-225).

Legs (all on the same scan sequence; every leg's final map must equal the CPU checker's, see `self_check`):
  value / ms_per_step    clouds resident in HBM (ufomap_map_insert_device), async=true  -- the headline
  host_pointer           ufomap_map_insert with a PAGEABLE host cloud (24 B/point over PCIe inside the timed region;
                         pinned staging, copy overlapped), async=true                  -- = value_incl_h2d
  host_pinned            the same with the cloud in caller-owned pinned memory (DMA straight from it)
  pointcloud2            ufomap_map_insert_pointcloud2: float32 x, y, z records (16 B/point) + pose, async=true
  server_loop            per scan: pointcloud2 ingest + integrate, setValueVolume(robot box), writeData(changed AABB)
  sync_latency           async=false: one scan at a time, nothing overlapped
  resident_static        round 1's figure: one scan re-integrated into a saturated map from a static pose
  other_configs          BASELINE configs C1, C5, C3 (insert depth 6 / 3 / 0): ms per scan, each map checked against the
                         reference's digest (tests/golden/digests.json) or the reference itself
At N > 1 every rank integrates its own moving sensor (pose (rank + i) mod 8): one call of ufomap_map_insert_batch per
step -- the ranks exchange their scans as bit grids over RCCL (one all-gather) and every replica of the map applies
all N scans in rank order with ONE walk of the tree ("scaling": "weak").

Prints ONE JSON line on rank 0: metric/value/unit = integrated rays/s (input points per second, whole job), plus
  roofline     -- dominant kernel (the ray walk k_fcast: one launch = one scan): algorithmic bytes per launch / HIP-event
                  duration vs 8 TB/s; `frac_rocprof` = the same with rocprofv3's average duration (profiles/);
                  `dominant_by_time` = the kernel with the largest total time; `traffic` = PMC bytes (profiles/)
  cpu_baseline -- the reference (oracle/_ref) or the oracle port on this box's host cores, same scan sequence
  pipeline     -- how many scans shared a walk of the tree, host time per scan
Exit code 3 when a leg's map differs from the CPU checker's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RES, MAX_RANGE, DEPTH = 0.16, 20.0, 0
P_BYTES = 24  # bytes per input point (3 x f64), SURVEY 8(d)
N_POSES = 8
MIN_TIMED_S = 0.5


def algorithmic_bytes(n_points, hit_codes, miss_codes, steps, levels=16):
    """B_scan = P*N + 16*S + 16*(U_h+U_f) + 40*sum_d U_d, counted for the exact input."""
    touched = np.union1d(hit_codes, miss_codes)
    sum_ud = 0
    k = touched
    for _ in range(1, levels + 1):
        k = np.unique(k >> np.uint64(3))
        sum_ud += len(k)
    terms = dict(points=P_BYTES * n_points, keys=16 * steps, cells=16 * (len(hit_codes) + len(miss_codes)), parents=40 * sum_ud)
    return sum(terms.values()), terms, sum_ud


def make_clouds():
    from ufomap_amd import scans
    return [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(N_POSES)]


def cpu_baseline(clouds, seq, budget_s=12.0):
    """Reference (or port) on the host cores: the same scan sequence into a fresh map, repeated while the budget lasts."""
    from oracle import OracleMap, available, build
    build("port")
    kind = "reference" if available("reference") else "port"
    per_scan, reps = [], 0
    t_start = time.perf_counter()
    while reps < 1 or time.perf_counter() - t_start < budget_s:
        m = OracleMap(RES, kind=kind)
        for p in seq:
            origin, xyz, _ = clouds[p]
            t0 = time.perf_counter()
            m.insert(origin, xyz, max_range=MAX_RANGE, depth=DEPTH, discrete=True)
            per_scan.append(time.perf_counter() - t0)
        reps += 1
    mean = float(np.mean(per_scan))
    n = clouds[0][1].shape[0]
    return dict(value=n / mean, unit="rays/s", cores=2 if kind == "reference" else 1, kind=kind,
                sample=f"{reps} x the bench's {len(seq)}-scan moving-sensor sequence (131072-pt scans) into a fresh map on the host "
                       f"(mean {mean * 1e3:.1f} ms/scan, median {float(np.median(per_scan)) * 1e3:.1f}, first {per_scan[0] * 1e3:.1f})",
                ms_per_scan_mean=mean * 1e3, ms_per_scan_median=float(np.median(per_scan)) * 1e3)


def live_traffic(kernels, timeout_s=150):
    """HBM bytes per launch of `kernels`, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, each in a run of its own with
    --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short run of this script's headline leg in a child process. Returns
    (bytes or None, per-kernel dict, note). The library hands over with events instead of gate kernels when it sees the counter
    collection (kernels are serialised across streams under it)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("UFOMAP_BENCH_CHILD"):
        return None, {}, "rocprofv3 not on PATH" if not exe else "child run"
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-self-check", "--only-headline",
           "--profile-kernels", "0", "--min-timed-s", "0.05"]
    env = dict(os.environ, UFOMAP_BENCH_CHILD="1", TMPDIR="/tmp")
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="ufo_pmc_", dir="/tmp")
            subprocess.run([exe, "-f", "csv", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--"] + cmd, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            acc = {}
            for f in files:
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != counter:
                        continue
                    k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("ufo::", "").strip().split("<")[0]
                    k = "k_fcast" if k in ("k_fcast2", "k_fcast3", "k_fcast4") else k  # (the later forms of the ray kernel: the library times them all as k_fcast)
                    a = acc.setdefault(k, [0, 0.0])
                    a[0] += 1
                    a[1] += float(r.get("Counter_Value") or 0)
            shutil.rmtree(d, ignore_errors=True)
            for k, (n, tot) in acc.items():
                per.setdefault(k, {})[counter] = tot / max(n, 1)
    except Exception as e:  # noqa: BLE001
        return None, {}, f"rocprofv3 --pmc failed: {e!r}"
    out = {}
    for k, v in per.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[k] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0  # (the guide's gfx950 correction: the read side doubled)
    vals = [out.get(k) for k in kernels]
    if not all(vals):
        return None, out, "the counters of " + ", ".join(k for k, v in zip(kernels, vals) if not v) + " were not collected"
    return float(sum(vals)), out, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, a pass each, over 25 scans of the headline leg in a child process"


def other_configs(device, big, only=None):
    """BASELINE configs C1, C5 and C3 (insert depth 6 / 3 / 0) on this GPU: sync calls, cloud resident in HBM, the fixture's scan
    sequence into a fresh map. The map after every scan is compared with the UNMODIFIED reference: its digest recorded in
    tests/golden/digests.json (C1, C5, C3 at depth 0: 85-170 s per scan on the CPU), or the reference run here (C3 at depth 6 / 3)."""
    import torch
    from ufomap_amd import OccupancyMap, OccupancyMapColor, scans
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util
    fixtures = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
    out = {}

    def gen(g, kw):
        kw = dict(kw)
        if "pose" in kw:
            kw["origin"] = scans.lidar_pose(kw.pop("pose"))
        return getattr(scans, g)(**kw)

    def settle_allocator(n_bytes):
        """After a leg has given tens of GB back, the runtime's next large hipMalloc can take hundreds of ms (409 / 628 ms measured
        inside the coloured frame's first scan, 1 ms on other runs): let it do that outside any timed call."""
        try:
            x = torch.empty(int(n_bytes), dtype=torch.uint8, device=f"cuda:{device}")
            x.zero_()
            torch.cuda.synchronize()
            del x
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001  (hygiene only)
            pass

    def run(label, params, seq, want, warm_reps, instrument=False):
        from ufomap_amd import capi
        params = dict(params)
        if instrument:
            settle_allocator(24 << 30)
        color = params.pop("color", False)
        m = (OccupancyMapColor if color else OccupancyMap)(device=device, **params)
        ms, ok = [], True
        allocs_fixture, allocs_warm = [], []
        last = None
        for k, (g, gkw, ikw) in enumerate(seq):
            origin, xyz, rgb = gen(g, gkw)
            d = torch.from_numpy(xyz).to(f"cuda:{device}")
            drgb = torch.from_numpy(rgb).to(f"cuda:{device}") if (rgb is not None and color) else None
            torch.cuda.synchronize()
            a0 = capi.alloc_counters()
            t0 = time.perf_counter()
            m.insert_device(origin, d.data_ptr(), drgb.data_ptr() if drgb is not None else None, xyz.shape[0], ikw.get("max_range", -1.0),
                            ikw.get("depth", 0), ikw.get("discrete", False))
            ms.append((time.perf_counter() - t0) * 1e3)
            a1 = capi.alloc_counters()
            allocs_fixture.append({k: a1[k] - a0[k] for k in a1})
            if want is not None:
                ok = ok and [str(v) for v in m.digest()] == want[k]
            last = (origin, d, drgb, xyz.shape[0], ikw)
        c = m.last_counts()
        st = m.stats()
        dig = tuple(m.digest())  # (of the fixture's scans: before the warm repetitions below)
        warm = []
        warm_checked = 0
        for rep in range(warm_reps):  # the last scan again into the now warm map (values saturate; what a static sensor costs)
            origin, d, drgb, n, ikw = last
            torch.cuda.synchronize()
            a0 = capi.alloc_counters()
            t0 = time.perf_counter()
            m.insert_device(origin, d.data_ptr(), drgb.data_ptr() if drgb is not None else None, n, ikw.get("max_range", -1.0), ikw.get("depth", 0),
                            ikw.get("discrete", False))
            warm.append((time.perf_counter() - t0) * 1e3)
            a1 = capi.alloc_counters()
            allocs_warm.append({k: a1[k] - a0[k] for k in a1})
            # (outside the timed call: the map after THIS warm scan against the reference's after as many scans of the frame, where
            # the fixture holds them -- round 4 timed warm scans whose results nobody had checked at full size)
            if want is not None and len(seq) + rep < len(want):
                ok = ok and [str(v) for v in m.digest()] == want[len(seq) + rep]
                warm_checked += 1
        kernels = None
        if instrument and last is not None:
            # one more warm repetition with HIP events around every launch (outside the timed ones): which kernel the time is in
            origin, d, drgb, n, ikw = last
            m.set_profiling(True)
            m.reset_kernel_times()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.insert_device(origin, d.data_ptr(), drgb.data_ptr() if drgb is not None else None, n, ikw.get("max_range", -1.0), ikw.get("depth", 0),
                            ikw.get("discrete", False))
            prof_ms = (time.perf_counter() - t0) * 1e3
            kt = m.kernel_times()
            m.set_profiling(False)
            if want is not None and len(seq) + warm_reps < len(want):
                ok = ok and [str(v) for v in m.digest()] == want[len(seq) + warm_reps]
                warm_checked += 1
            kernels = dict(ms_with_events=round(prof_ms, 3),
                           per_kernel_ms={k: round(v["total_ms"], 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]) if v["launches"]},
                           launches={k: v["launches"] for k, v in kt.items() if v["launches"]})
        out[label] = dict(points=c["points"], rays=c["rays"], dda_steps=c["steps"], ms_per_scan_fixture=[round(v, 4) for v in ms],
                          ms_warm_median=(float(np.median(warm)) if warm else None), digest_ok=bool(ok), checked_against=("tests/golden/digests.json (unmodified reference)" if want is not None else None),
                          live_blocks=st["inner_nodes"], leaves=st["leaf_nodes"], table_bytes=st["bytes"],
                          fast_path_scans=int(m.debug()[61]), scans=len(seq) + warm_reps, warm_scans_digest_checked=warm_checked)
        if instrument:
            # (a fresh map's first scan allocates the table and the scratch arrays: hipMalloc / hipFree of 10-13 GB take 1 ms on one
            # run and 400 ms on another -- the runtime's business, counted by the library and taken out here)
            out[label]["ms_per_scan_fixture_minus_alloc_host_time"] = [round(v - a["host_ns"] * 1e-6, 4) for v, a in zip(ms, allocs_fixture)]
            out[label].update(ms_warm_all=[round(v, 3) for v in warm], ms_warm_min=float(min(warm)) if warm else None,
                              device_allocs_in_fixture_calls=allocs_fixture, device_allocs_in_warm_calls=allocs_warm, warm_kernels=kernels)
        return dig

    def c3_pipelined(device, fx):
        """The frame again and again with async=true: a call returns with its tree update enqueued, the next call casts its rays
        meanwhile (the volume path's two halves overlap across scans). Map after the last scan against the reference's."""
        n_async = len(fx["steps"]) - 1
        if n_async < 2:
            return None
        origin, xyz, _ = gen(*fx["scans"][0][:2])
        ikw = fx["scans"][0][2]
        d = torch.from_numpy(xyz).to(f"cuda:{device}")
        m = OccupancyMap(device=device, **fx["params"])
        # (every hand-over set allocates and clears brick grids of its own when it is first used -- 4.5 GB each for this frame: three
        # untimed scans touch both sets the pipeline alternates between, then the map is cleared and the counted sequence starts)
        for k in range(3):
            m.insert_device(origin, d.data_ptr(), None, xyz.shape[0], ikw.get("max_range", -1.0), 0, True, False, 0, k > 0)
        m.insertPointCloudWait()
        m.clear()
        m.insert_device(origin, d.data_ptr(), None, xyz.shape[0], ikw.get("max_range", -1.0), 0, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_async):
            m.insert_device(origin, d.data_ptr(), None, xyz.shape[0], ikw.get("max_range", -1.0), 0, True, False, 0, True)
        m.insertPointCloudWait()
        ms = (time.perf_counter() - t0) * 1e3 / n_async
        ok = [str(v) for v in m.digest()] == fx["steps"][-1]["digest"]
        return dict(ms_per_scan=ms, scans=n_async, digest_ok=bool(ok), note="async=true calls back to back after one synchronous scan into a fresh map; "
                    "the final map compared with the unmodified reference's after the same number of scans")

    for label, name, reps in (("C1_lidar16cm_continuous", "c1_full", 10), ("C5_lidar8cm_colour", "c5_colour_8cm", 10)) + (
            (("C3_rgbd2mm_depth0", "c3_depth0_full", 5), ("C3_rgbd2mm_colour_depth0", "c3_colour_full", 5)) if big else ()):
        if only and label not in only:
            continue
        if name not in fixtures:
            continue
        fx = fixtures[name]
        c3 = label == "C3_rgbd2mm_depth0"
        if label == "C3_rgbd2mm_colour_depth0":
            # the reference's only published figure (README.md:10-11): a COLOURED map at 2 mm, "real-time 2 Hz" = 500 ms per frame.
            # The same frame with colours on the volume path (round 5; round 4: the general path); fresh + warm, the first warm
            # repetitions compared with the reference's map after as many scans
            run(label, fx["params"], fx["scans"][:1], [s["digest"] for s in fx["steps"]], reps, instrument=True)
            out[label]["reference_published_ms"] = 500.0
            torch.cuda.empty_cache()
            continue
        # (C3 at insert depth 0: the fixture holds the SAME frame seven times -- the first scan is timed into a fresh map, the others
        # are the warm repetitions, every one of them compared with the reference's map after as many scans)
        run(label, fx["params"], fx["scans"][:1] if c3 else fx["scans"], [s["digest"] for s in fx["steps"]], reps, instrument=c3)
        if c3:
            out[label]["pipelined"] = c3_pipelined(device, fx)
        if label == "C3_rgbd2mm_depth0":
            # the one bandwidth-bound configuration: SURVEY 8d's B_scan = 24 N + 16 S + 16 (touched voxels) + 40 (touched node blocks),
            # with the fresh map's leaves / inner nodes as the touched voxels / blocks of its single scan
            c = out[label]
            b = 24 * c["points"] + 16 * c["dda_steps"] + 16 * c["leaves"] + 40 * c["live_blocks"]
            c["algorithmic_bytes"] = b
            c["hbm_roofline_frac_fresh_map"] = b / (c["ms_per_scan_fixture"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS
            c["hbm_roofline_frac_warm_map"] = b / (c["ms_warm_median"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        torch.cuda.empty_cache()
    # C3 at insert depth 6 / 3: the reference itself is the checker (80-400 ms per scan on the CPU)
    from oracle import OracleMap, available, build
    build("port")
    kind = "reference" if available("reference") else "port"
    go, gx, _ = scans.rgbd()
    for depth in (6, 3):
        label = f"C3_rgbd2mm_depth{depth}"
        if only and label not in only:
            continue
        seq = [("rgbd", {}, dict(max_range=5.0, depth=depth, discrete=True))] * 2
        dig = run(label, dict(resolution=0.002), seq, None, 6)
        o = OracleMap(0.002, kind=kind)
        for _ in range(2):
            o.insert(go, gx, max_range=5.0, depth=depth, discrete=True)
        out[label]["digest_ok"] = bool(dig == tuple(golden_util.dump_digest(o.leaves(True), o.inner())))
        out[label]["checked_against"] = f"the {kind} build run here on the same two scans"
        del o
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--python-exchange", action="store_true", help="N > 1: exchange the update lists with torch.distributed instead of the C ABI's own RCCL call")
    # (the driver runs `--steps 20 --warmup 5`; the defaults are the same since round 5 -- rounds 2-4 ran 40 / 8 here, and what was
    # read as "the driver's box is 10 % slower" was the pipeline's fill and drain, ~200 us per timed region, spread over 20 steps
    # instead of 40: the same box gives 0.0496 ms at K = 20, 0.0443 at 40, 0.0410 at 100, 0.0395 at 400)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true")
    ap.add_argument("--force-batch", action="store_true", help="run the N>1 code path (scan/exchange/apply) even with one rank")
    ap.add_argument("--profile-kernels", type=int, default=1, help="extra leg with every kernel bracketed by HIP events (roofline)")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S)
    ap.add_argument("--only-headline", action="store_true", help="skip the extra legs (profiling runs)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/pmc_latest.json instead of two rocprofv3 --pmc passes in this run")
    ap.add_argument("--big", type=int, default=1, help="other_configs: include C3 at insert depth 0 (2 mm, 3.4e8 leaves, ~11 GB of node table)")
    ap.add_argument("--no-other-configs", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    # The timed loops are Python: no cyclic garbage collection inside them (as `timeit` does it). With torch imported a full
    # collection stops this thread for ~35 ms -- 800 scans' worth -- once in a few thousand calls; the library's own caller,
    # a C++ node, has no such pauses (scripts/dev/dev_ab.py shows the per-repetition spread with and without).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    batch_mode = world > 1 or args.force_batch
    # RCCL prints a version banner on stdout: keep stdout clean for the ONE JSON line (banner -> stderr)
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if batch_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from ufomap_amd import OccupancyMap, PointCloud

    K, W = args.steps, args.warmup
    clouds = make_clouds()
    n_pts = clouds[0][1].shape[0]
    d_clouds = [torch.from_numpy(c[1]).to(dev) for c in clouds]  # resident in HBM before any timed region
    pose_of = lambda i, r=rank: (r + i) % N_POSES  # noqa: E731  (step i of rank r)
    seq = [pose_of(i) for i in range(W + K)]

    def barrier():
        torch.cuda.synchronize()
        if batch_mode:
            dist.barrier()

    def run_leg(m, step_fn, min_timed_s, prepare=None, max_reps=4000):
        """W warm-up + exactly K timed steps from a cleared map, repeated until min_timed_s of timed steps."""
        dts = []
        while True:
            m.insertPointCloudWait()
            m.clear()
            if prepare:
                prepare()
            for i in range(W):
                step_fn(i)
            m.insertPointCloudWait()
            barrier()
            t0 = time.perf_counter()
            for i in range(W, W + K):
                step_fn(i)
            m.insertPointCloudWait()
            barrier()
            dt = time.perf_counter() - t0
            if batch_mode:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            dts.append(dt)
            done = sum(dts) >= min_timed_s or len(dts) >= max_reps
            if batch_mode:  # every rank takes the same decision
                flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
                dist.broadcast(flag, 0)
                done = bool(flag.item())
            if done:
                return dts

    def leg_summary(dts, scans_per_step=1):
        tot = float(sum(dts))
        return dict(rays_per_s=n_pts * scans_per_step * K * len(dts) / tot, ms_per_step=tot / (K * len(dts)) * 1e3,
                    ms_per_step_median_rep=float(np.median(dts)) / K * 1e3, repeats=len(dts), timed_region_s=tot)

    m = OccupancyMap(RES, device=local_rank)
    if batch_mode:
        from ufomap_amd import dist as udist
        # the batch step is ONE call of the C ABI (ufomap_map_insert_batch: scan, RCCL all-gather, apply); torch.distributed
        # only carries the communicator id, the barriers and the max over ranks of this script. --python-exchange selects
        # the same protocol written with torch collectives (ufomap_amd/dist.py: BatchIntegrator), which is also what the
        # gloo tests drive on CPU.
        batch_impl = "c_abi:ufomap_map_insert_batch"
        if args.python_exchange:
            batch = udist.BatchIntegrator(m, dist.group.WORLD, dev)
            batch_impl = "python:torch.distributed all_gather + ufomap_map_apply_keys_batch"
        else:
            batch = udist.CBatchIntegrator(m, local_rank, dist.group.WORLD)

        def step_resident(i):
            p = pose_of(i)
            batch.integrate(clouds[p][0], d_clouds[p].data_ptr(), n_pts, MAX_RANGE, DEPTH, discrete=True)
    else:
        def step_resident(i):
            p = pose_of(i)
            m.insert_device(clouds[p][0], d_clouds[p].data_ptr(), None, n_pts, MAX_RANGE, DEPTH, discrete=True, async_=True)

    # ---- headline leg -----------------------------------------------------------------------------------------
    dbg0 = m.debug()
    dts = run_leg(m, step_resident, args.min_timed_s)
    head = leg_summary(dts, world)
    dbg1 = m.debug()
    n_scans_leg = max(1, dbg1[61] - dbg0[61])
    pipeline = dict(fast_path_scans=dbg1[61] - dbg0[61], tree_walks=dbg1[60] - dbg0[60],
                    scans_per_walk=(dbg1[59] - dbg0[59]) / max(1, dbg1[60] - dbg0[60]), predicted_grid_repeats=dbg1[63] - dbg0[63],
                    gate_timeouts=dbg1[58] - dbg0[58],
                    host_us_per_scan={k: (dbg1[52 + j] - dbg0[52 + j]) / n_scans_leg * 1e-3 for j, k in enumerate(("scan_half_enqueue", "tree_update_enqueue", "join", "call_total"))},
                    note="headline leg (warm-up scans included): scans that queue up behind the map stream share ONE walk of the tree")
    digests = {"resident": m.digest()}
    mem_stats = m.stats()
    final_dump = (m.leaves(True), m.inner()) if rank == 0 else None

    extra = {}
    ktimes = {}
    if not batch_mode and not args.only_headline:
        # ---- the call the reference's server makes: host pointer in, H2D inside the timed region ----------------
        host_clouds = [PointCloud(c[1].copy()) for c in clouds]  # pageable

        def step_host(i):
            p = pose_of(i)
            m.insertPointCloudDiscrete(clouds[p][0], host_clouds[p], MAX_RANGE, DEPTH, False, 0, True)
        extra["host_pointer"] = dict(leg_summary(run_leg(m, step_host, args.min_timed_s)),
                                     note="ufomap_map_insert, pageable 24 B/point cloud: host memcpy into pinned staging + async H2D inside the timed region; async=true")
        digests["host_pointer"] = m.digest()
        pinned = [torch.from_numpy(c[1].copy()).pin_memory() for c in clouds]
        pinned_clouds = [PointCloud(t.numpy()) for t in pinned]

        def step_pinned(i):
            p = pose_of(i)
            m.insertPointCloudDiscrete(clouds[p][0], pinned_clouds[p], MAX_RANGE, DEPTH, False, 0, True)
        extra["host_pinned"] = dict(leg_summary(run_leg(m, step_pinned, args.min_timed_s)),
                                    note="ufomap_map_insert, cloud in caller-owned pinned memory: DMA straight from it, the call returns when the copy is done")
        digests["host_pinned"] = m.digest()

        def step_sync(i):
            p = pose_of(i)
            m.insert_device(clouds[p][0], d_clouds[p].data_ptr(), None, n_pts, MAX_RANGE, DEPTH, discrete=True, async_=False)
        extra["sync_latency"] = dict(leg_summary(run_leg(m, step_sync, min(args.min_timed_s, 0.25))), note="async=false, HBM-resident clouds")
        digests["sync"] = m.digest()

        # ---- one step of the multi-GPU batch path with ONE rank (the N = 1 line carries what `--force-batch`
        # measures): ufomap_map_insert_batch through the real RCCL -- scan, pack, all-gather of one slot, the walk -- so that the
        # driver's run shows what a batch step costs beside a pipelined single-GPU scan, whether or not a multi-GPU node is at hand
        try:
            from ufomap_amd import Comm
            comm1 = Comm(Comm.unique_id(), 1, 0, local_rank)
            m.set_option("async_apply", 1)  # (as ufomap_amd.dist.CBatchIntegrator does: a step returns with its walk enqueued)

            def step_batch1(i):
                p = pose_of(i)
                m.insert_batch(comm1, clouds[p][0], d_clouds[p].data_ptr(), n_pts, MAX_RANGE, DEPTH, True)
            extra["batch_step_n1"] = dict(leg_summary(run_leg(m, step_batch1, min(args.min_timed_s, 0.25))), rccl_ranks=1, counters=comm1.counters(),
                                          note="ufomap_map_insert_batch with a communicator of ONE rank over the real librccl (the N > 1 code path: bit-grid steps, "
                                               "one all-gather per step, the walk enqueued by the host); compare with ms_per_step of the pipelined single-GPU path")
            digests["batch_n1"] = m.digest()
            m.insertPointCloudWait()
            m.set_option("async_apply", 0)
            comm1.close()
        except Exception as e:  # (librccl missing on the box: reported, the other legs stand)
            extra["batch_step_n1"] = dict(error=repr(e))

        # ---- the raw records of a PointCloud2 (float32 x, y, z, pad: 16 B/point) + the sensor's pose, host memory --------
        rec = []
        ident = np.array([1.0, 0.0, 0.0, 0.0])
        for origin, xyz, _ in clouds:
            b = np.zeros((n_pts, 4), np.float32)
            b[:, :3] = (xyz - origin[None, :]).astype(np.float32)  # sensor frame
            rec.append(np.ascontiguousarray(b).view(np.uint8).reshape(-1))

        def step_pc2(i):
            p = pose_of(i)
            m.insertPointCloud2(clouds[p][0], ident, rec[p], 16, (0, 4, 8), None, MAX_RANGE, DEPTH, True, False, 0, True)
        extra["pointcloud2"] = dict(leg_summary(run_leg(m, step_pc2, args.min_timed_s)),
                                    note="ufomap_map_insert_pointcloud2: float32 records (16 B/point over PCIe) + pose; rosToUfo + transform fused into the first kernel; async=true")
        digests_pc2 = {"pointcloud2": m.digest()}

        # ---- the server's per-message sequence (server.cpp:114-225): ingest + integrate, clear the robot's volume, serialise
        # the part of the map that changed (writeData of the change AABB, what ufoToMsg publishes) ---------------------------
        robot = np.array([0.5, 0.5, 0.75])
        pub = dict(bytes=0, msgs=0)

        def step_server(i):
            p = pose_of(i)
            m.insertPointCloud2(clouds[p][0], ident, rec[p], 16, (0, 4, 8), None, MAX_RANGE, DEPTH, True, False, 0, True)
            m.setValueVolume(clouds[p][0] - robot, clouds[p][0] + robot, m.getClampingThresMin(), 0)
            mn, mx = m.minmax_change()
            m.resetMinMaxChangeDetection()
            data, _ = m.write_ex(aabb=(mn, mx), compress=False, min_depth=0, header=False)
            pub["bytes"] += len(data)
            pub["msgs"] += 1
        dts_srv = run_leg(m, step_server, min(args.min_timed_s, 0.25), prepare=m.resetMinMaxChangeDetection)
        extra["server_loop"] = dict(leg_summary(dts_srv), bytes_per_publish=pub["bytes"] / max(1, pub["msgs"]),
                                    note="per scan: pointcloud2 ingest + integrate, setValueVolume(robot box, clamping_thres_min), writeData(change AABB) "
                                         "+ resetMinMaxChangeDetection (server.cpp:114-225)")
        digests_pc2["server_loop"] = m.digest()
        m.resetMinMaxChangeDetection()

        # ---- the headline loop driven from C++ through the C ABI (examples/bench_loop.cpp): the library without the Python
        # interpreter and ctypes between two calls -- what a C++ caller such as the reference's server gets ------------------
        try:
            import subprocess
            import tempfile
            from ufomap_amd import build as hipbuild
            exe = hipbuild.build_bench_loop(verbose=False)
            with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as tf:
                tf.write(np.array([N_POSES, n_pts], np.uint64).tobytes())
                for p in range(N_POSES):
                    tf.write(np.asarray(clouds[p][0], np.float64).tobytes())
                    tf.write(np.ascontiguousarray(clouds[p][1], np.float64).tobytes())
                path = tf.name
            pr = subprocess.run([exe, path, str(W), str(K), str(args.min_timed_s), str(local_rank)], capture_output=True, text=True, timeout=300)
            os.unlink(path)
            if pr.returncode == 0:
                cx = json.loads(pr.stdout.strip().splitlines()[-1])
                extra["host_cxx"] = dict(rays_per_s=cx["rays_per_s"], ms_per_step=cx["ms_per_step"], repeats=cx["repeats"], timed_region_s=cx["timed_region_s"],
                                         scans_per_walk=cx["fast_path_scans"] / max(1, cx["tree_walks"]), gate_timeouts=cx["gate_timeouts"],
                                         host_us_per_call=cx.get("host_us_per_call"), host_us_scan_half_enqueue=cx.get("host_us_scan_half_enqueue"),
                                         host_us_map_half_enqueue=cx.get("host_us_map_half_enqueue"), host_us_join=cx.get("host_us_join"),
                                         device_timeline=dict(cx.get("timeline", {}), note="one more repetition of the C++ loop with the hand-over kernels recording the device "
                                                              "clock (ufomap_map_timeline): medians. The period of the pipeline is the larger of the scan stream's and the host's "
                                                              "call rate: a box whose ms_per_step is higher shows WHERE here -- ray_kernel_and_launches_us (device: clock, "
                                                              "queue mapping) or published_to_next_gate_us / host_us_per_call (host: launch cost)"),
                                         note="the same W + K sequence and repetitions as `value`, driven by examples/bench_loop.cpp (a C++ loop over "
                                              "ufomap_map_insert_device, async) in a process of its own: the library without the Python loop")
                digests["host_cxx"] = tuple(int(v) for v in cx["digest"])
            else:
                extra["host_cxx"] = dict(error=f"rc {pr.returncode}: {pr.stderr.strip()[-300:]}")
        except Exception as e:  # noqa: BLE001  (the leg is a diagnostic: it must not take the bench line down)
            extra["host_cxx"] = dict(error=repr(e))

        # ---- round 1's figure: one scan re-integrated from a static pose into a saturated map -------------------
        ms = OccupancyMap(RES, device=local_rank)
        d0, o0 = d_clouds[0], clouds[0][0]
        for _ in range(max(W, 12)):
            ms.insert_device(o0, d0.data_ptr(), None, n_pts, MAX_RANGE, DEPTH, discrete=True, async_=True)
        ms.insertPointCloudWait()
        torch.cuda.synchronize()
        reps_s = max(K, int(0.25 / 1.5e-4))
        t0 = time.perf_counter()
        for _ in range(reps_s):
            ms.insert_device(o0, d0.data_ptr(), None, n_pts, MAX_RANGE, DEPTH, discrete=True, async_=True)
        ms.insertPointCloudWait()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        extra["resident_static"] = dict(rays_per_s=n_pts * reps_s / dt_s, ms_per_step=dt_s / reps_s * 1e3, steps=reps_s,
                                        note="steady-state best case: same scan, static pose, saturated map (round 1's headline conditions)")
        del ms

    # ---- the other BASELINE configurations: ms per scan, every map checked ----------------------------------------------
    other = None
    if not batch_mode and not args.only_headline and not args.no_other_configs and rank == 0:
        other = other_configs(local_rank, bool(args.big))

    # ---- per-kernel HIP events (roofline leg): sync calls so that a kernel's events do not straddle overlapped work --
    dt_ev = None
    if args.profile_kernels and not batch_mode:
        m.reset_kernel_times()
        m.set_profiling(True)
        dt_ev = run_leg(m, step_resident, 0.0, max_reps=3)
        m.set_profiling(False)
        ktimes = m.kernel_times()
        n_prof_steps = (W + K) * len(dt_ev)
        digests["events"] = m.digest()

    # ---- counts of the exact inputs (per pose, fresh map): algorithmic bytes, rays cast, DDA steps -------------------
    counts, alg = [], []
    if rank == 0:
        mc = OccupancyMap(RES, device=local_rank)
        for p in sorted(set(seq[W:])):
            mc.clear()
            mc.insert_device(clouds[p][0], d_clouds[p].data_ptr(), None, n_pts, MAX_RANGE, DEPTH, discrete=True)
            c = mc.last_counts()
            hits, misses = mc.last_hits(), mc.last_misses()
            b, terms, sum_ud = algorithmic_bytes(n_pts, hits, misses, c["steps"])
            counts.append(dict(pose=p, rays=c["rays"], steps=c["steps"], hits=len(hits), miss_cells=len(misses), sum_U_d=sum_ud))
            alg.append((b, terms))
        del mc

    # ---- self-check: every leg ends on the map the CPU checker builds from the same W + K scans ----------------------
    self_check = None
    if rank == 0 and not args.no_self_check:
        from oracle import OracleMap, available, build
        build("port")
        kind = "reference" if available("reference") else "port"
        t0 = time.perf_counter()
        o = OracleMap(RES, kind=kind)
        n_scans = 0
        for i in range(W + K):
            for r in range(world):
                origin, xyz, _ = clouds[pose_of(i, r)]
                o.insert(origin, xyz, max_range=MAX_RANGE, depth=DEPTH, discrete=True)
                n_scans += 1
        ol, oi = o.leaves(True), o.inner()
        same = all(np.array_equal(a, b) for a, b in zip(final_dump[0], ol)) and all(np.array_equal(a, b) for a, b in zip(final_dump[1], oi))
        legs_equal = len(set(digests.values())) == 1
        self_check = dict(checker=kind, scans=n_scans, leaves=int(len(ol[0])), inner=int(len(oi[0])), map_equals_checker=bool(same),
                          legs_agree=bool(legs_equal), legs=sorted(digests), seconds=round(time.perf_counter() - t0, 1))
        if not batch_mode and not args.only_headline:
            # the PointCloud2 legs: the reference fed through ITS ingest (rosToUfo + transform of the same records), then -- for the
            # server loop -- its setValueVolume and resetMinMaxChangeDetection per scan
            import oracle as _or
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import golden_util
            o2, o3 = OracleMap(RES, kind=kind), OracleMap(RES, kind=kind)
            for i in range(W + K):
                p = pose_of(i)
                xyz32, _ = _or.ingest(rec[p], 16, (0, 4, 8), None, ident, clouds[p][0], kind)
                for om in (o2, o3):
                    om.insert(clouds[p][0], xyz32, max_range=MAX_RANGE, depth=DEPTH, discrete=True)
                o3.setValueVolume(clouds[p][0] - robot, clouds[p][0] + robot, o3.clamping_thres()[0], 0)
            ok2 = tuple(digests_pc2["pointcloud2"]) == tuple(golden_util.dump_digest(o2.leaves(True), o2.inner()))
            ok3 = tuple(digests_pc2["server_loop"]) == tuple(golden_util.dump_digest(o3.leaves(True), o3.inner()))
            self_check.update(pointcloud2_equals_checker=bool(ok2), server_loop_equals_checker=bool(ok3))
            same = same and ok2 and ok3
            self_check["map_equals_checker"] = bool(same)
        if other:
            self_check["other_configs_ok"] = all(v["digest_ok"] for v in other.values())

    if rank == 0:
        value = head["rays_per_s"]
        mean_rays = float(np.mean([c["rays"] for c in counts]))
        mean_steps = float(np.mean([c["steps"] for c in counts]))
        b_scan = float(np.mean([a[0] for a in alg]))
        terms = {k: float(np.mean([a[1][k] for a in alg])) for k in alg[0][1]}
        roof = None
        kern_ms = {k: (v["total_ms"] / max(v["launches"], 1)) for k, v in ktimes.items() if v["launches"]}
        per_step_ms = {k: v["total_ms"] / n_prof_steps for k, v in ktimes.items() if v["launches"]} if ktimes else {}
        if kern_ms:
            # The dominant kernel by algorithmic bytes is the ray walk: it carries the 16*S term of B_scan (82 %). It is
            # one launch (k_cast: set-up + segment queue + walk) plus the slab merge for LiDAR-sized scans, or
            # set-up + walk + merge for the other grid sizes; the durations of whatever ran add up.
            walkers = ("k_fcast", "k_cast", "k_walk", "k_dda_seg", "k_dda")  # fast path first: it runs all steady-state scans
            dom = next(k for k in walkers if k in kern_ms)
            # (k_fcast: one launch = one scan. The slab merge k_fmerge is part of the tree walk now -- one launch for all the scans
            # the walk takes -- and is not counted into the ray walk's duration any more.)
            group = [dom] if dom == "k_fcast" else [k for k in ("k_ray_setup",) + walkers[1:] + ("k_merge_slabs",) if k in kern_ms]
            group = [k for k in group if k in kern_ms]
            share = P_BYTES * mean_rays + 16 * mean_steps
            dur_s = sum(kern_ms[k] for k in group) * 1e-3  # average launch durations of the kernels that make up one ray walk
            achieved = share / dur_s / 1e9
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    vals = [next((v.get("hbm_bytes_per_launch") for kk, v in pj.items() if kk == k or kk.startswith(k + "<") or (k == "k_fcast" and kk.startswith(("k_fcast2", "k_fcast3", "k_fcast4")))), None) for k in group]
                    traffic = sum(v for v in vals if v) if any(vals) else None
                    import hashlib
                    traffic_src = ("profiles/pmc_latest.json (sha256 " + hashlib.sha256(open(pmc, "rb").read()).hexdigest()[:16] +
                                   "): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/profile_gpu.sh -- a builder-run file, not measured in this run")
                except Exception:
                    traffic = None
            traffic_live, traffic_live_kernels, traffic_live_note = (None, {}, "switched off") if args.no_live_traffic else live_traffic(group)
            if traffic_live is not None:
                traffic, traffic_src = traffic_live, traffic_live_note
            by_time = max(per_step_ms, key=per_step_ms.get)
            frac_rocprof, rocprof_src = None, None
            for tag in ("r06", "r05", "r04", "r03", "r02"):
                f = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
                if os.path.exists(f):
                    import csv
                    import hashlib
                    rows = {("k_fcast" if r["kernel"].split("<")[0] in ("k_fcast2", "k_fcast3", "k_fcast4") else r["kernel"].split("<")[0]): float(r["avg_ns"]) for r in csv.DictReader(open(f)) if r.get("avg_ns")}
                    if all(k in rows for k in group):
                        frac_rocprof = share / (sum(rows[k] for k in group) * 1e-9) / 1e9 / HBM_PEAK_GBS
                        rocprof_src = f"profiles/{tag}_kernel_stats.csv (sha256 {hashlib.sha256(open(f, 'rb').read()).hexdigest()[:16]}): rocprofv3 --kernel-trace --stats of this command"
                    break
            roof = dict(bound="hbm", kernel="+".join(group), achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        frac_rocprof=frac_rocprof, frac_rocprof_source=rocprof_src, traffic=traffic, traffic_source=traffic_src,
                        traffic_live_note=traffic_live_note, traffic_per_kernel={k: round(v) for k, v in traffic_live_kernels.items() if k in per_step_ms},
                        # (round 2's definition of the ray walk -- the slab merge counted with it -- for like-for-like comparisons)
                        with_fmerge=(dict(avg_launch_us=(kern_ms[dom] + kern_ms["k_fmerge"]) * 1e3, frac=share / ((kern_ms[dom] + kern_ms["k_fmerge"]) * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                     if "k_fmerge" in kern_ms and dom == "k_fcast" else None),
                        algorithmic_bytes_per_launch=share, avg_launch_us=dur_s * 1e6,
                        walk_kernel_only=dict(avg_launch_us=kern_ms[dom] * 1e3, achieved_GBs=share / (kern_ms[dom] * 1e-3) / 1e9,
                                              frac=share / (kern_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS),
                        dominant_by_time=dict(kernel=by_time, us_per_step=per_step_ms[by_time] * 1e3, avg_launch_us=kern_ms[by_time] * 1e3),
                        whole_scan=dict(algorithmic_bytes=b_scan, terms=terms, achieved_GBs=b_scan / (head["ms_per_step"] * 1e-3) / 1e9,
                                        frac=b_scan / (head["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS),
                        kernels_us_per_step={k: round(v * 1e3, 2) for k, v in sorted(per_step_ms.items(), key=lambda kv: -kv[1])},
                        launches_per_step=round(sum(v["launches"] for v in ktimes.values()) / n_prof_steps, 2))
        out = {
            "metric": "integrated rays/sec (input points per second, insertPointCloudDiscrete, 16 cm leaf, 20 m max-range)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": head["ms_per_step"], "ms_per_step_median_rep": head["ms_per_step_median_rep"],
            "repeats": head["repeats"], "timed_region_s": head["timed_region_s"],
            "rays_cast_per_s": value / n_pts * mean_rays, "dda_steps_per_s": value / n_pts * mean_steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 ray casting / u64 Morton keys / f32 log-odds", "data": "synthetic",
            "config": {"workload": ("configs[1]: synthetic 64-beam LiDAR scans, 131072 pts each, 16 cm leaf, 20 m max-range, discrete integrator + "
                                    "free-space raycast; moving sensor (pose i mod 8 of configs[3], 3 m apart), fresh map per repetition, clouds resident in HBM, async=true")
                       if not batch_mode else
                       "configs[3]: batch of N concurrent 131072-pt LiDAR scans per step (moving sensors), 16 cm leaf, one scan per GPU, ONE RCCL all-gather of the scans as bit grids, every replica applies all N in rank order with one walk of the tree",
                       "points_per_scan": n_pts, "rays_cast_mean": mean_rays, "dda_steps_mean": mean_steps, "per_pose": counts,
                       "leaf_m": RES, "max_range_m": MAX_RANGE, "depth_levels": 16, "parallelism": f"scan-per-gpu x{world}",
                       **({"batch_impl": batch_impl, "rccl_ranks": world} if batch_mode else {})},
            **({"rccl_ranks": world} if batch_mode else {}),
            "roofline": roof, "self_check": self_check, "pipeline": pipeline,
            "memory": {"table_bytes": mem_stats["bytes"], "live_blocks": mem_stats["inner_nodes"], "leaves": mem_stats["leaf_nodes"],
                       "bytes_per_live_block": mem_stats["bytes_per_block"],
                       "note": "node table as allocated after the headline leg's last repetition: tile-major since round 4 -- 73 slots of 68 B (round 5: one array per field; round 4: 80 B) per depth-3 tile behind a directory, "
                               "sized for every tile of the ray grid's hull being new. A LiDAR map is sparse inside its tiles (surfaces; free space is pruned away): ~10 live blocks "
                               "per 73-slot group, which is what bytes_per_live_block shows; the dense 2 mm RGB-D map (other_configs.C3_rgbd2mm_depth0) holds 47 live blocks per group"},
        }
        out.update(extra)
        if "host_pointer" in extra:
            out["value_incl_h2d"] = extra["host_pointer"]["rays_per_s"]
            out["value_definitions"] = ("value: clouds resident in HBM when the timed region starts (the task's measurement rule); value_incl_h2d: the call the "
                                        "reference's server makes, pageable host cloud in, 24 B/point over PCIe inside the timed region (SURVEY.md 8d's definition)")
        if other:
            out["other_configs"] = other
        if dt_ev:
            out["ms_per_step_with_events"] = float(sum(dt_ev)) / (K * len(dt_ev)) * 1e3
        if not batch_mode and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(clouds, seq)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)  # RCCL's banner sits in the C stdio buffer
        except Exception:
            pass
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    bad = bool(self_check) and not (self_check["map_equals_checker"] and self_check["legs_agree"] and self_check.get("other_configs_ok", True))
    bad = bad or (rank == 0 and not batch_mode and pipeline["gate_timeouts"] > 0)  # a stream hand-over timed out inside a timed region
    if batch_mode:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.stderr.write("bench.py: SELF-CHECK FAILED -- a leg's final map differs from the CPU checker's\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
