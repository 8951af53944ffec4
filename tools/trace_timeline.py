"""Timeline of one forward from a rocprofv3 kernel_trace.csv (tools/trace_forward.py markers): every launch in start order with
its queue, start offset, duration and the idle gap before it on its queue.  Usage: trace_timeline.py trace.csv [min_gap_us]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
sel = rows[marks[-2] + 1:marks[-1]]
t0 = int(sel[0]["Start_Timestamp"])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else -1.0
last_end, qn = {}, {}
for r in sel:
    q = qn.setdefault(r.get("Queue_Id", "0"), len(qn))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    if gap >= min_gap:
        print(f"q{q} t={(s - t0) / 1e3:8.1f}  dur={(e - s) / 1e3:7.1f}  gap={gap:7.1f}  grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']})  {name[:60]}")
