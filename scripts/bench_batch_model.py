"""One GPU: the pieces of a multi-GPU batch step, measured separately, and the modelled step time for
N = 1, 2, 4, 8 ranks. NOT a scaling curve: the collective is priced from the list sizes and the xGMI ring bandwidth, the
rest is measured here (ray casting of one scan; one walk of the tree for N lists)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans

ENTRY = 16
clouds = []
for p in range(8):
    origin, xyz, _ = scans.lidar64(origin=scans.lidar_pose(p), seed=100 + p)
    clouds.append((origin, torch.from_numpy(xyz).cuda(), xyz.shape[0]))


def lists_for(m, poses):
    out = []
    for p in poses:
        o, d, n = clouds[p]
        info = m.scan_keys(o, d.data_ptr(), n, 20.0, 0, True)
        buf = torch.empty((info.n_hit + info.n_miss) * ENTRY, dtype=torch.uint8, device="cuda")
        m.get_keys(buf.data_ptr(), buf.numel() // ENTRY, info)
        out.append((buf, info))
    return out


res = {"workload": "C4: 131072-pt LiDAR scans, 16 cm, 20 m, discrete, pose k of the 8-pose set per rank k; warm map (all 8 poses integrated before)"}
m = OccupancyMap(0.16)
# warm map
for p in range(8):
    o, d, n = clouds[p]
    m.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True)
# scan_keys alone
ts = []
for rep in range(30):
    o, d, n = clouds[rep % 8]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = m.scan_keys(o, d.data_ptr(), n, 20.0, 0, True)
    ts.append(time.perf_counter() - t0)
res["scan_keys_ms"] = float(np.median(ts[5:])) * 1e3
res["list_bytes_mean"] = float(np.mean([(k.n_hit + k.n_miss) * ENTRY for _, k in lists_for(m, range(8))]))
# apply N lists with one walk
for N in (1, 2, 4, 8):
    ls = lists_for(m, range(N))
    ts = []
    for rep in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.apply_keys_batch([b.data_ptr() for b, _ in ls], [k for _, k in ls])
        m.insertPointCloudWait()
        ts.append(time.perf_counter() - t0)
    res[f"apply_{N}_lists_ms"] = float(np.median(ts[5:])) * 1e3
# single-GPU integration of one scan (what N = 1 costs without any list)
ts = []
for rep in range(30):
    o, d, n = clouds[rep % 8]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.insert_device(o, d.data_ptr(), None, n, 20.0, 0, discrete=True)
    ts.append(time.perf_counter() - t0)
res["insert_sync_ms"] = float(np.median(ts[5:])) * 1e3
# model: ring all-gather of N slots over xGMI, 1 link direction ~ 50 GB/s effective per hop, latency ~ 20 us per collective
link_GBs, lat_us = 50.0, 20.0
model = {}
for N in (1, 2, 4, 8):
    gather_ms = 0.0 if N == 1 else (lat_us * 1e-3 + (N - 1) * res["list_bytes_mean"] / (link_GBs * 1e9) * 1e3)
    step = res["scan_keys_ms"] + gather_ms + res[f"apply_{N}_lists_ms"]
    overlapped = max(res["scan_keys_ms"] + gather_ms, res[f"apply_{N}_lists_ms"])
    model[N] = dict(gather_ms=gather_ms, step_serial_ms=step, step_overlapped_ms=overlapped, scans_per_s_serial=N / step * 1e3,
                    scans_per_s_overlapped=N / overlapped * 1e3)
res["model"] = model
res["model_assumptions"] = f"ring all-gather: {lat_us} us + (N-1) x list bytes / {link_GBs} GB/s; serial = scan + gather + apply, overlapped = max(scan + gather, apply) (async_apply: the next batch's scan runs while the tree is updated)"
print(json.dumps(res))
