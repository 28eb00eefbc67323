"""Developer aid: per-queue timeline statistics from a rocprofv3 --kernel-trace CSV (the steady-state part of a run):
average duration per kernel, busy fraction and launch-to-launch gaps per hardware queue, kernels per second.
usage: python scripts/trace_timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
# the steady-state part: between the launches of the ray kernel at `skip` and at 95 % of its launches (a bench run ends with
# self-check exports and starts with warm-up, neither of which is the pipeline)
fc = [int(r["Start_Timestamp"]) for r in rows if "k_fcast" in r["Kernel_Name"]]
if len(fc) > 20:
    lo, hi = fc[int(len(fc) * skip)], fc[int(len(fc) * 0.95)]
else:
    lo, hi = t0 + (t1 - t0) * skip, t1
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) <= hi]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) * 1e-3
per_q = defaultdict(list)
dur = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("ufo::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per_q[r["Queue_Id"]].append((s, e, name))
    dur[name].append((e - s) * 1e-3)
print(f"span {span:.0f} us, {len(rows)} kernels")
for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {name:28s} n={len(d):6d} avg {sum(d) / len(d):8.2f} us  total {sum(d) / span * 100:5.1f}% of span")
for q, ev in sorted(per_q.items()):
    busy = sum(e - s for s, e, _ in ev) * 1e-3
    gaps = [(ev[i + 1][0] - ev[i][1]) * 1e-3 for i in range(len(ev) - 1)]
    names = defaultdict(int)
    for _, _, n in ev:
        names[n] += 1
    small = [g for g in gaps if g < 50]
    print(f"queue {q}: {len(ev)} kernels, busy {busy / span * 100:.1f}%, median gap {sorted(gaps)[len(gaps) // 2] if gaps else 0:.2f} us, mean gap<50us {sum(small) / max(1, len(small)):.2f} us; "
          + ", ".join(f"{n}x{c}" for n, c in sorted(names.items(), key=lambda kv: -kv[1])[:6]))
# periods of the ray kernel and of the walks
for key in ("k_fcast", "k_ftail", "k_tile"):
    st = sorted(int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"])
    if len(st) > 2:
        d = sorted((st[i + 1] - st[i]) * 1e-3 for i in range(len(st) - 1))
        print(f"{key}: start-to-start median {d[len(d) // 2]:.1f} us, p10 {d[len(d) // 10]:.1f}, p90 {d[9 * len(d) // 10]:.1f}")
