"""C3 at insert depth 0, asynchronous calls back to back (the volume path's halves overlapping across scans): for a kernel trace.
   rocprofv3 --kernel-trace -f csv -d out -o t -- python scripts/dev/dev_vol_pipe.py ; python scripts/dev/dev_vol_pipe.py --show out/.../t_kernel_trace.csv"""
import sys, os, time
if len(sys.argv) > 2 and sys.argv[1] == "--show":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tiles = [i for i, r in enumerate(rows) if "k_tile" in r["Kernel_Name"]]
    # (the script runs 2 + 8 asynchronous scans, then 2 + 8 synchronous ones: the last four walks are synchronous scans, the four
    # that end ten walks before the end are asynchronous ones in the steady state)
    asy = len(sys.argv) > 3 and sys.argv[3] == "async"
    lo = (tiles[-15] if len(tiles) >= 15 else 0) if asy else (tiles[-4] if len(tiles) >= 4 else 0)
    hi = (tiles[-11] + 6 if len(tiles) >= 15 else len(rows)) if asy else len(rows)
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ufo::", "")[:40]
        s, e = (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - t0) * 1e-3
        if e - s > 8 or "k_tile" in name or "k_vwalk" in name:
            print(f"q{r['Queue_Id']:>3} {s:10.1f} {e:10.1f} {e - s:9.1f} us  {name}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ufomap_amd import OccupancyMap, scans
go, gx, _ = scans.rgbd()
d = torch.from_numpy(gx).cuda()
m = OccupancyMap(0.002)
for k in range(4):
    m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True, False, 0, k > 0)
m.insertPointCloudWait()
for lds in [int(a) for a in sys.argv[1:]] or [0]:
    m.set_option("vol_walk_lds", lds)
    res = {}
    for asy in (True, False):
        for _ in range(2):
            m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True, False, 0, asy)
        m.insertPointCloudWait()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            m.insert_device(go, d.data_ptr(), None, gx.shape[0], 5.0, 0, True, False, 0, asy)
        m.insertPointCloudWait()
        res["async" if asy else "sync"] = round((time.perf_counter() - t0) * 1e3 / 8, 3)
    print("vol_walk_lds", lds, "ms per scan over 8 calls", res, flush=True)
