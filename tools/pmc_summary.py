"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass).
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [<per-dispatch out.csv>]
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-byte read requests at 64 bytes); values in KB.
The optional fourth file lists the dispatches of the LAST marked region (tools/trace_forward.py brackets one forward with two
cumsum kernels) one per row in dispatch order, so that tools/pmc_to_json.py can attribute them to ABI calls by order."""
import csv, sys, collections, re


def rows_of(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def short(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")


def load(rows):
    agg = collections.OrderedDict()
    for r in rows:
        a = agg.setdefault((short(r["Kernel_Name"]), r["Grid_Size"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def marked(rows):
    marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
    return rows[marks[-2] + 1:marks[-1]] if len(marks) >= 2 else rows


fr, wr = rows_of(sys.argv[1], "FETCH_SIZE"), rows_of(sys.argv[2], "WRITE_SIZE")
f, w = load(fr), load(wr)
with open(sys.argv[3], "w") as o:
    o.write("kernel,grid_threads,launches,fetch_mb_per_launch_corrected_x2,write_mb_per_launch,total_mb_per_launch\n")
    rows = []
    for key, (n, kb) in f.items():
        wn, wkb = w.get(key, [n, 0.0])
        fm, wm = 2.0 * kb / n / 1024.0, wkb / max(wn, 1) / 1024.0
        rows.append((fm * n + wm * n, key, n, fm, wm))
    for tot, (name, grid), n, fm, wm in sorted(rows, reverse=True):
        o.write(f"\"{name}\",{grid},{n},{fm:.2f},{wm:.2f},{fm + wm:.2f}\n")
if len(sys.argv) > 4:
    # the k-th dispatch of a (kernel, grid) in the fetch pass and in the write pass is the same launch of the same program
    fm_, wm_ = marked(fr), marked(wr)
    wq = collections.defaultdict(collections.deque)
    for r in wm_:
        wq[(short(r["Kernel_Name"]), r["Grid_Size"])].append(float(r["Counter_Value"]))
    with open(sys.argv[4], "w") as o:
        o.write("order,kernel,grid_threads,fetch_mb_corrected_x2,write_mb\n")
        for i, r in enumerate(fm_):
            key = (short(r["Kernel_Name"]), r["Grid_Size"])
            wv = wq[key].popleft() if wq[key] else float("nan")
            o.write(f"{i},\"{key[0]}\",{key[1]},{2.0 * float(r['Counter_Value']) / 1024.0:.3f},{wv / 1024.0:.3f}\n")
