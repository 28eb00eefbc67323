"""One GPU: what a step of ufomap_map_insert_batch costs a rank when N ranks take part -- MEASURED on one MI355X with a
stand-in for RCCL that fills all N slots of the all-gather with this rank's own slot (tests/cpp/rccl_shim.cpp,
UFOMAP_SHIM_REPLICATE): the scan half, the packing, N slots landing in the receive buffer, and ONE walk of the tree for N
scans run exactly as they would on rank 0 of an N-GPU node; only the wire is missing, and it is priced separately (ring
all-gather over xGMI). NOT a scaling curve -- the driver measures that when it has an 8-GPU node."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shim_src, shim = os.path.join(ROOT, "tests", "cpp", "rccl_shim.cpp"), os.path.join(ROOT, "tests", "cpp", "librccl_shim.so")
if not os.path.exists(shim) or os.path.getmtime(shim) < os.path.getmtime(shim_src):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", shim_src, "-o", shim, "-lrt"], check=True)
os.environ["UFOMAP_RCCL_LIB"] = shim
os.environ["UFOMAP_SHIM_REPLICATE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ufomap_amd import OccupancyMap, scans  # noqa: E402
from ufomap_amd.occupancy_map import Comm  # noqa: E402

K, W = 40, 8
clouds = [scans.lidar64(origin=scans.lidar_pose(s), seed=100 + s) for s in range(8)]
d_clouds = [torch.from_numpy(c[1]).cuda() for c in clouds]
n_pts = clouds[0][1].shape[0]
res = {"workload": "C4: 131072-pt LiDAR scans, 16 cm, 20 m, discrete; step i integrates the scan of pose i mod 8 on every (emulated) rank; "
                   "fresh map per repetition, clouds resident in HBM, async_apply",
       "what": "ms per STEP on one rank when N ranks take part (scan half + pack + N slots gathered + one walk for N scans), wire excluded"}
for N in (1, 2, 4, 8):
    m = OccupancyMap(0.16)
    m.set_option("async_apply", 1)
    if os.environ.get("GATHER_STREAM"):
        m.set_option("gather_stream", 1)  # (A/B: the all-gather on a stream of its own)
    comm = Comm(Comm.unique_id(), N, 0, 0)
    dts = []
    for rep in range(40):
        m.insertPointCloudWait()
        m.clear()
        for i in range(W):
            m.insert_batch(comm, clouds[i % 8][0], d_clouds[i % 8].data_ptr(), n_pts, 20.0, 0, True)
        m.insertPointCloudWait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            m.insert_batch(comm, clouds[i % 8][0], d_clouds[i % 8].data_ptr(), n_pts, 20.0, 0, True)
        m.insertPointCloudWait()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    cnt = comm.counters()
    st = comm.stats()
    res[f"step_ms_N{N}"] = float(np.median(dts[3:])) / K * 1e3
    res[f"fast_steps_N{N}"] = cnt["fast_steps"]
    res[f"repeated_steps_N{N}"] = cnt["repeated_steps"]
    m.insertPointCloudWait()
    comm.close()
    del m
# the honest base of an efficiency: what ONE GPU does alone with the same scans -- the pipelined single-GPU path
# (ufomap_map_insert_device, async), same sequence, same run -- not N = 1 of the batch path
m = OccupancyMap(0.16)
dts = []
for rep in range(40):
    m.insertPointCloudWait()
    m.clear()
    for i in range(W):
        m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, True, async_=True)
    m.insertPointCloudWait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        m.insert_device(clouds[i % 8][0], d_clouds[i % 8].data_ptr(), None, n_pts, 20.0, 0, True, async_=True)
    m.insertPointCloudWait()
    torch.cuda.synchronize()
    dts.append(time.perf_counter() - t0)
res["single_gpu_pipelined_ms_per_scan"] = float(np.median(dts[3:])) / K * 1e3
del m
# the exchange slot of the fast-path form: 2 KiB + two bit grids of the ranks' common ray grid (the 8 poses' hull: ~95 KB each)
slot_bytes = 2048 + 2 * 97000
link_GBs, lat_us = 50.0, 20.0
model = {}
for N in (1, 2, 4, 8):
    wire_ms = 0.0 if N == 1 else (lat_us * 1e-3 + (N - 1) * slot_bytes / (link_GBs * 1e9) * 1e3)
    step = res[f"step_ms_N{N}"]
    # The gather sits on the scan stream behind the scan half; the walk of the step before runs on the map stream meanwhile.
    # Two bounds: the wire entirely hidden behind that walk (the measured step), and entirely exposed (measured step + wire).
    model[N] = dict(wire_ms=wire_ms, step_ms_wire_hidden=step, step_ms_wire_exposed=step + wire_ms,
                    scans_per_s_wire_hidden=N / step * 1e3, scans_per_s_wire_exposed=N / (step + wire_ms) * 1e3)
one_gpu = 1e3 / res["single_gpu_pipelined_ms_per_scan"]  # scans/s of one GPU integrating on its own
for N in model:
    # against N x ONE pipelined GPU (what N maps on N GPUs would integrate without any collective) ...
    model[N]["efficiency_wire_hidden"] = model[N]["scans_per_s_wire_hidden"] / (N * one_gpu)
    model[N]["efficiency_wire_exposed"] = model[N]["scans_per_s_wire_exposed"] / (N * one_gpu)
    # ... and, for comparison with rounds 2 / 3, against N x (N = 1 of the batch path)
    model[N]["efficiency_vs_batch_N1_wire_hidden"] = model[N]["scans_per_s_wire_hidden"] / (N * model[1]["scans_per_s_wire_hidden"])
    model[N]["efficiency_vs_batch_N1_wire_exposed"] = model[N]["scans_per_s_wire_exposed"] / (N * model[1]["scans_per_s_wire_exposed"])
res["model"] = model
res["model_assumptions"] = (f"ring all-gather over xGMI: {lat_us} us + (N-1) x {slot_bytes} B / {link_GBs} GB/s per link direction; efficiency = scans/s at N / "
                            "(N x scans/s of ONE GPU on the pipelined single-GPU path, measured in this run); measured part: everything but the wire, on one MI355X playing rank 0 of N")
print(json.dumps(res))
