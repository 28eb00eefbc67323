"""Condense rocprofv3 output (scripts/profile_gpu.sh) into the small files kept under profiles/."""
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""  # "c3": the passes over scripts/dev/dev_c3d0.py (directories c3_stats, c3_pmc_fetch, c3_pmc_write)
if sub:
    tag = tag + "_" + sub
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "gpurun_out", "profiles_" + sys.argv[2])
os.makedirs(prof, exist_ok=True)


def short(name):
    # "void ufo::k_dda<false, 0>(ufo::MapGeom, ...)" -> "k_dda<false, 0>"
    n = name.split("(")[0].replace("void ", "").replace("ufo::", "").strip()
    return n


def find(pattern):
    hits = glob.glob(os.path.join(out_dir, "**", pattern), recursive=True)
    hits = [h for h in hits if (("/" + sub + "_") in h) == bool(sub)]
    return sorted(hits)[0] if hits else None


summary = {}
stats = find("*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
        for r in rows:
            w.writerow([short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")])
            summary.setdefault(short(r.get("Name", "")), {}).update(calls=int(r.get("Calls") or 0), avg_ns=float(r.get("AverageNs") or 0))

for counter, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = find(f"{key}*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r.get("Kernel_Name", ""))
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r.get("Counter_Value") or 0)
    for k, (n, tot) in acc.items():
        summary.setdefault(k, {})[counter + "_per_launch"] = tot / max(n, 1)
        summary[k][counter + "_launches"] = n

# HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are in KiB-like units of
# 1024 B... the guide's formula is hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, and on gfx950 FETCH_SIZE
# reads 1/2 of the bytes of a wide coalesced stream, so the read side is doubled (upper estimate for
# narrow accesses, which are uncalibrated).
for k, v in summary.items():
    if "FETCH_SIZE_per_launch" in v or "WRITE_SIZE_per_launch" in v:
        fs, ws = v.get("FETCH_SIZE_per_launch", 0.0), v.get("WRITE_SIZE_per_launch", 0.0)
        v["hbm_bytes_per_launch_raw"] = (fs + ws) * 1024.0
        v["hbm_bytes_per_launch"] = (2.0 * fs + ws) * 1024.0
json.dump(summary, open(os.path.join(prof, f"{tag}_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k.startswith("k_")}, indent=1)[:3000])
