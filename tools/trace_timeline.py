"""Launch-by-launch timeline of one forward from a rocprofv3 kernel_trace.csv around tools/trace_forward.py (the launches
between the two cumsum markers): start offset, duration, idle gap before it on its own queue, queue, kernel, grid."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
sel = rows[marks[-2] + 1:marks[-1]]
t0 = int(sel[0]["Start_Timestamp"])
last_end = {}
qids = {}
for r in sel:
    q = qids.setdefault(r.get("Queue_Id", "0"), len(qids))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{q}  {name[:60]:60s} ({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']})")
