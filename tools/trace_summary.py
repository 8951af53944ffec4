"""Summarise a rocprofv3 kernel_trace.csv produced around tools/trace_forward.py: launches between the two cumsum markers."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
lo, hi = marks[-2], marks[-1]
sel = rows[lo + 1:hi]
t0 = int(sel[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in sel)
print(f"{len(sel)} launches, span {(t1 - t0) / 1e3:.1f} us, sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3:.1f} us")
agg = collections.OrderedDict()
for r in sel:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for (name, gx, gy, gz), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f"{d:9.1f} us  x{n:<3d} {d / n:8.1f} each  grid=({gx},{gy},{gz})  {name[:70]}")
