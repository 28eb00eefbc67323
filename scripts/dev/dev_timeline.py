"""Developer aid (GPU box): where the streams of the steady-state pipeline wait. The bench's headline loop with option
"tstamps": device clock at every hand-over (include/ufomap_hip.h: ufomap_map_timeline), no tracing tool involved.
usage: python scripts/dev/dev_timeline.py ["opt=val,opt=val" ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, scans  # noqa: E402

K, W = 400, 16
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d_clouds = [torch.from_numpy(c[1]).cuda() for c in clouds]
n_pts = clouds[0][1].shape[0]

for spec in sys.argv[1:] or [""]:
    m = OccupancyMap(0.16)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            m.set_option(k, int(v))
    m.set_option("tstamps", 1)
    for rep in range(3):
        m.insertPointCloudWait()
        m.clear()
        for i in range(W):
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, discrete=True, async_=True)
        m.insertPointCloudWait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, discrete=True, async_=True)
        m.insertPointCloudWait()
        dt = time.perf_counter() - t0
    ts, newest = m.timeline()
    idx = np.arange(newest - K + 40, newest - 10) % 4096
    r = ts[idx].astype(np.int64)
    us = lambda a: a * 0.01  # noqa: E731  (100 MHz)
    per = us(np.diff(r[:, 3]))
    print(f"--- {spec or 'defaults'}: {dt / K * 1e6:.1f} us/scan (host clock), scan-half period median {np.median(per):.1f} us (p10 {np.percentile(per, 10):.1f}, p90 {np.percentile(per, 90):.1f})")
    def stat(name, a):
        a = us(a)
        print(f"    {name:58s} median {np.median(a):7.1f}  mean {a.mean():7.1f}  p10 {np.percentile(a, 10):7.1f}  p90 {np.percentile(a, 90):7.1f}")
    stat("gate wait (scan stream idle for the first-point pass)", r[:, 2] - r[:, 1])
    stat("first-point pass done -> gate open (prep ahead if > 0)", r[:, 2] - r[:, 0])
    stat("gate open -> scan half published (k_fcast + launches)", r[:, 3] - r[:, 2])
    stat("published(i) -> gate entered(i+1)", r[1:, 1] - r[:-1, 3])
    stat("first-point pass: signal(i) - signal(i-1)", np.diff(r[:, 0]))
    own = r[r[:, 7] > 0]
    stat("claim wait (map stream idle for a scan half)", own[:, 5] - own[:, 4])
    stat("claim done -> tree update done (k_fmerge, k_tile, k_ftail)", own[:, 6] - own[:, 5])
    stat("scans per walk", own[:, 7] * 100)
    stat("walk period (tree update done, consecutive walks)", np.diff(own[:, 6]))
    stat("tree update done -> next claim entered", own[1:, 4] - own[:-1, 6])
    stat("scan published -> its walk's tree update done (latency)", own[:, 6] - own[:, 3])
    del m
