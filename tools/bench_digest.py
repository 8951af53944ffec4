"""Short digest of a bench.py JSON line.  usage: bench_digest.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "per_rank_s", "rank_spread") if k in d})
r = d["roofline"]
print("dominant:", r["kernel_class"], "frac", r["frac"], "avg launch us", round(r["avg_launch_us"], 1), r["kernel"], "traffic", r.get("traffic"))
for c in d["roofline_table"][:10]:
    print(f"  {c['class']:45s} share {c['share']:.3f}  frac {c.get('frac')}  {c['kernel_ms']:.0f} ms")
g = d.get("g1_ball_query_and_grouping")
if g:
    print("g1:", g["grouping_gather"], g["ball_query"], "pair", g["pair_hbm_frac"])
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], "speedup", d.get("speedup_vs_cpu_baseline"), "c1", d["cpu_baseline"].get("c1_full"))
