"""Developer A/B (GPU box): latency of a synchronous insert (HBM-resident cloud, moving sensor) under different options.
usage: python scripts/dev/dev_sync.py "solo=0" "solo=1" ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, scans  # noqa: E402

clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d_clouds = [torch.from_numpy(c[1]).cuda() for c in clouds]
n_pts = clouds[0][1].shape[0]
for spec in sys.argv[1:] or [""]:
    m = OccupancyMap(0.16)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            m.set_option(k, int(v))
    ts = []
    for rep in range(6):
        m.clear()
        for i in range(8):
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, discrete=True, async_=False)
        for i in range(8, 48):
            t0 = time.perf_counter()
            m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, discrete=True, async_=False)
            ts.append(time.perf_counter() - t0)
    ts = np.array(ts[40:]) * 1e6
    d = m.debug()
    st = d[30:39]
    print(f"{spec:30s} sync insert: median {np.median(ts):.1f} us, mean {ts.mean():.1f}, p10 {np.percentile(ts, 10):.1f}, p90 {np.percentile(ts, 90):.1f}   "
          f"k_fcast wg0 stamps {[round((x - st[0]) / 100.0, 1) for x in st]}  k_ftail {round((d[19] - d[10]) / 100.0, 1)} us; fcast start -> ftail end {round((d[19] - d[30]) / 100.0, 1)} us",
          flush=True)
    del m
