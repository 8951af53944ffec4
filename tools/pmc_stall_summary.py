"""Summarise the four rocprofv3 --pmc passes of tools/pmc_stalls.sh (gpurun_out/stalls_{1..4}.csv): per kernel (name, grid)
the SQ wave-cycle buckets (parked on s_waitcnt / barrier, issue stall, issuing), L2 hit rate, L1->L2 request ratio, TA busy.
usage: pmc_stall_summary.py <dir with stalls_1..4.csv> [top n]"""
import csv, collections, re, sys, os

def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, cnt, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        key = (re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (key, r["Dispatch_Id"]) not in seen:
            seen.add((key, r["Dispatch_Id"])); cnt[key] += 1
            dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, cnt, dur

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
a1, c1, d1 = load(os.path.join(d, "stalls_1.csv")); a2, _, _ = load(os.path.join(d, "stalls_2.csv"))
a3, _, _ = load(os.path.join(d, "stalls_3.csv")); a4, _, _ = load(os.path.join(d, "stalls_4.csv"))
print("three PC2 forwards (B=16, N=4096) under rocprofv3 --pmc; durations are inflated by the counter collection")
print(f"{'kernel':44s} {'grid':>9s} {'n':>3s} {'us':>7s} | parked% stall% issue% | L2hit% L1->L2/L1acc | TAbusy(Mcyc)")
for key in sorted(a1, key=lambda k: -a1[k].get("SQ_WAVE_CYCLES", 0))[:top]:
    v, n = a1[key], c1[key]
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    v2, v3, v4 = a2.get(key, {}), a3.get(key, {}), a4.get(key, {})
    req = v2.get("TCC_REQ_sum", 0) or 1
    acc = v3.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) or 1
    print(f"{key[0][:44]:44s} {key[1]:>9s} {n:3d} {d1[key] / n:7.1f} | {100 * v.get('SQ_WAIT_ANY', 0) / wc:6.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:6.1f} "
          f"{100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1f} | {100 * v2.get('TCC_HIT_sum', 0) / req:5.1f} {v3.get('TCP_TCC_READ_REQ_sum', 0) / acc:10.2f} | "
          f"{v4.get('TA_TA_BUSY_sum', 0) / n / 1e6:8.1f}")
