"""Round 6: the host-pointer legs (pageable / pinned clouds) under option stage_thread, with the library's own per-call host time:
python scripts/dev/dev_host.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, PointCloud, scans
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
n = clouds[0][1].shape[0]
pinned = [torch.from_numpy(c[1].copy()).pin_memory() for c in clouds]
for kind in ("pageable", "pinned"):
    for st in (0, 1):
        m = OccupancyMap(0.16)
        m.set_option("stage_thread", st)
        for o in sys.argv[1:]:
            k, v = o.split("=")
            m.set_option(k, int(v))
        pcs = [PointCloud(c[1].copy()) for c in clouds] if kind == "pageable" else [PointCloud(t.numpy()) for t in pinned]
        for rep in range(3):
            m.insertPointCloudWait(); m.clear()
            for i in range(16):
                m.insertPointCloudDiscrete(clouds[i % 8][0], pcs[i % 8], 20.0, 0, False, 0, True)
            m.insertPointCloudWait(); torch.cuda.synchronize()
            d0 = m.debug(); t0 = time.perf_counter()
            for i in range(16, 216):
                m.insertPointCloudDiscrete(clouds[i % 8][0], pcs[i % 8], 20.0, 0, False, 0, True)
            t1 = time.perf_counter()
            m.insertPointCloudWait(); dt = time.perf_counter() - t0
            d1 = m.debug()
        print(f"{kind:9s} stage_thread={st}: {dt / 200 * 1e6:7.1f} us per scan; calls return after {(t1 - t0) / 200 * 1e6:6.1f} us; inside doInsert {(d1[55] - d0[55]) / 200 * 1e-3:6.1f} us "
              f"(scan half {(d1[52] - d0[52]) / 200 * 1e-3:5.1f}, slot {(d1[53] - d0[53]) / 200 * 1e-3:5.1f}, join {(d1[54] - d0[54]) / 200 * 1e-3:5.1f})", flush=True)
