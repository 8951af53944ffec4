"""Short view of a rocprofv3 *kernel_stats.csv: name cut at the first '(' + calls / average / min / max (us).  python tools/kernel_stats_short.py <csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.1f}  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f} us")
