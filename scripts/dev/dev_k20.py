"""The bench's headline leg at the driver's settings (25-scan repetitions from a cleared map: 5 warm-up scans, 20 timed, the region ends
with a wait) under library options: python scripts/dev/dev_k20.py "opt=val,opt=val" ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d = [torch.from_numpy(c[1]).cuda() for c in clouds]
n = clouds[0][1].shape[0]
K, W = 20, 5
for spec in sys.argv[1:] or [""]:
    m = OccupancyMap(0.16)
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            m.set_option(k, int(v))
    tot, reps = 0.0, 0
    for rep in range(260):
        m.insertPointCloudWait(); m.clear()
        for i in range(W):
            m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True, False, 0, True)
        m.insertPointCloudWait(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(W, W + K):
            m.insert_device(clouds[i % 8][0], d[i % 8].data_ptr(), None, n, 20.0, 0, True, False, 0, True)
        m.insertPointCloudWait(); torch.cuda.synchronize()
        if rep >= 10:
            tot += time.perf_counter() - t0; reps += 1
    print(spec or "defaults", "ms per step", round(tot / (K * reps) * 1e3, 5), "G rays/s", round(n * K * reps / tot / 1e9, 3), flush=True)
