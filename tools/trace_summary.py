"""Summarise a rocprofv3 kernel_trace.csv produced around tools/trace_forward.py: launches between the two cumsum markers."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
lo, hi = marks[-2], marks[-1]
sel = rows[lo + 1:hi]
t0 = int(sel[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in sel)
print(f"{len(sel)} launches, span {(t1 - t0) / 1e3:.1f} us, sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e3:.1f} us")
# idle gaps per hardware queue (= stream): time between the end of one kernel and the start of the next on the same queue
byq = collections.defaultdict(list)
for r in sel:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    gaps = [max(0, iv[i + 1][0] - iv[i][1]) for i in range(len(iv) - 1)]
    busy = sum(e - s0 for s0, e in iv)
    if gaps:
        gs = sorted(gaps)
        print(f"queue {q}: {len(iv)} launches, busy {busy / 1e3:.1f} us, gaps total {sum(gaps) / 1e3:.1f} us, median gap {gs[len(gs) // 2] / 1e3:.2f} us, "
              f"p90 {gs[int(len(gs) * 0.9)] / 1e3:.2f} us, span {(iv[-1][1] - iv[0][0]) / 1e3:.1f} us")
agg = collections.OrderedDict()
for r in sel:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for (name, gx, gy, gz), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f"{d:9.1f} us  x{n:<3d} {d / n:8.1f} each  grid=({gx},{gy},{gz})  {name[:70]}")
