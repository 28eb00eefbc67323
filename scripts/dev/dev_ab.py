"""Developer A/B (GPU box): the bench's headline loop (moving sensor, HBM-resident clouds, async) under different options.
usage: python scripts/dev/dev_ab.py "batch_max=1" "batch_max=8" "batch_max=8,cast_wgs=128" ...
Prints ms/scan, pipeline counters and (second pass, events on) average kernel times per configuration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, OccupancyMapColor, scans  # noqa: E402

RES = float(os.environ.get("RES", "0.16"))       # leaf size (0.08: the ray grid lives in HBM)
COLOR = bool(int(os.environ.get("COLOR", "0")))  # colour map + coloured clouds

K, W = 40, 8
if os.environ.get("NOGC"):
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s, colored=COLOR) for s in range(8)]
d_clouds = [torch.from_numpy(c[1]).cuda() for c in clouds]
d_rgb = [torch.from_numpy(c[2]).cuda() if COLOR else None for c in clouds]
n_pts = clouds[0][1].shape[0]


def run(m, reps):
    dts = []
    for _ in range(reps):
        m.insertPointCloudWait()
        m.clear()
        for i in range(W):
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), d_rgb[i % 8].data_ptr() if COLOR else None, n_pts, 20.0, 0, discrete=True, async_=True)
        m.insertPointCloudWait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), d_rgb[i % 8].data_ptr() if COLOR else None, n_pts, 20.0, 0, discrete=True, async_=True)
        m.insertPointCloudWait()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    return dts


ref_digest = None
for spec in sys.argv[1:] or [""]:
    m = (OccupancyMapColor if COLOR else OccupancyMap)(RES)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            m.set_option(k, int(v))
    run(m, 3)
    d0 = m.debug()
    dts = run(m, int(os.environ.get("REPS", "60")))
    d1 = m.debug()
    dig = m.digest()
    if ref_digest is None:
        ref_digest = dig
    ns = max(1, d1[61] - d0[61])
    srt = np.sort(dts) / K * 1e3
    print("   rep ms/scan: min %.4f p25 %.4f p50 %.4f p75 %.4f p90 %.4f max %.4f; reps slower than 1.3 x median: %d of %d" % (
        srt[0], srt[len(srt) // 4], srt[len(srt) // 2], srt[3 * len(srt) // 4], srt[9 * len(srt) // 10], srt[-1], int((srt > 1.3 * srt[len(srt) // 2]).sum()), len(srt)))
    line = f"{spec:45s} ms/scan {np.sum(dts) / (K * len(dts)) * 1e3:.4f} (median rep {np.median(dts) / K * 1e3:.4f})  scans/walk {(d1[59] - d0[59]) / max(1, d1[60] - d0[60]):.2f}" \
           f"  host us/scan {(d1[55] - d0[55]) / ns * 1e-3:.1f} (scan-half enq {(d1[52] - d0[52]) / ns * 1e-3:.1f}, slot enq {(d1[53] - d0[53]) / ns * 1e-3:.1f}, join {(d1[54] - d0[54]) / ns * 1e-3:.1f})  redo {d1[63] - d0[63]} gate_to {d1[58]} same_map {dig == ref_digest}"
    m.reset_kernel_times()
    m.set_profiling(True)
    run(m, 2)
    m.set_profiling(False)
    kt = m.kernel_times()
    line += "  | " + " ".join(f"{k}={v['total_ms'] / max(1, v['launches']) * 1e3:.1f}us" for k, v in sorted(kt.items()) if v["launches"])
    print(line, flush=True)
    del m
