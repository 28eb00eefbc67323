"""Timings of the other BASELINE configs (parity-test cases, not bench lines): one MI355X, HBM-resident
cloud, median of repeats of ufomap_map_insert_device into a warm map (first = fresh map)."""
import json, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ufomap_amd import OccupancyMap, OccupancyMapColor, scans


def run(name, cls, res, origin, xyz, rgb, max_range, depth, discrete, reps):
    d = torch.from_numpy(xyz).cuda()
    drgb = torch.from_numpy(rgb).cuda() if rgb is not None else None
    m = cls(res)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.insert_device(origin, d.data_ptr(), drgb.data_ptr() if drgb is not None else None, xyz.shape[0], max_range, depth, discrete)
        ts.append(time.perf_counter() - t0)
    top = None
    if "--kernels" in sys.argv:
        m.reset_kernel_times(); m.set_profiling(True)
        for _ in range(5):
            m.insert_device(origin, d.data_ptr(), drgb.data_ptr() if drgb is not None else None, xyz.shape[0], max_range, depth, discrete)
        m.set_profiling(False)
        kt = m.kernel_times()
        top = {k: round(v["total_ms"] / 5 * 1e3, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:7]}
    c = m.last_counts()
    st = m.stats()
    out = dict(config=name, points=xyz.shape[0], rays=c["rays"], steps=c["steps"], hits=c["hits"], ms_fresh=ts[0] * 1e3,
               ms_warm_median=float(np.median(ts[1:])) * 1e3 if reps > 1 else None,
               rays_per_s_warm=xyz.shape[0] / float(np.median(ts[1:])) if reps > 1 else None, live_blocks=st["inner_nodes"], leaves=st["leaf_nodes"],
               table_bytes=st["bytes"], kernels_us=top)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    lo, lx, lc = scans.lidar64(colored=True)
    go, gx, _ = scans.rgbd()
    run("C1 lidar 16cm continuous", OccupancyMap, 0.16, lo, lx, None, 20.0, 0, False, 20)
    run("C2 lidar 16cm discrete", OccupancyMap, 0.16, lo, lx, None, 20.0, 0, True, 20)
    run("C5 lidar 8cm colour discrete", OccupancyMapColor, 0.08, lo, lx, lc, 20.0, 0, True, 20)
    run("C3 rgbd 2mm depth6", OccupancyMap, 0.002, go, gx, None, 5.0, 6, True, 10)
    run("C3 rgbd 2mm depth3", OccupancyMap, 0.002, go, gx, None, 5.0, 3, True, 10)
    if "--big" in sys.argv:
        run("C3 rgbd 2mm depth0", OccupancyMap, 0.002, go, gx, None, 5.0, 0, True, 3)
