"""Developer aid: the kernels of a window of a rocprofv3 --kernel-trace CSV in start order, one column per hardware queue:
python scripts/dev/trace_seq.py <kernel_trace.csv> [fraction_of_run=0.8] [n_kernels=60] [name_filter_for_anchor=k_pack_slot]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 60
anchor = sys.argv[4] if len(sys.argv) > 4 else "k_pack_slot"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
an = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = an[int(len(an) * frac)] if an else int(len(rows) * frac)
win = rows[i0:i0 + nk]
qs = sorted({r["Queue_Id"] for r in win})
t0 = int(win[0]["Start_Timestamp"])
print("start_us  end_us   " + "".join(f"queue {q:<22s}" for q in qs))
for r in win:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ufo::", "")[:24]
    s, e = (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - t0) * 1e-3
    col = qs.index(r["Queue_Id"])
    print(f"{s:8.1f} {e:8.1f}  " + " " * (28 * col) + f"{name} ({e - s:.1f})")
