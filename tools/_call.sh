cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6
rm -rf /tmp/trs; timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o st -- python3 $R/tools/trace_step.py > /dev/null 2>&1
F=$(find /tmp/trs -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/r6/c15_timeline.txt
python3 $R/tools/trace_step_summary.py $R/gpurun_out/r6/c15_timeline.txt | cut -c1-300 | head -10
python3 - <<'P'
import re, os
rows=[]
for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r6/c15_timeline.txt'):
    m = re.match(r'q(\d+) t=\s*([\d.]+)\s+dur=\s*([\d.]+)\s+gap=\s*(-?[\d.]+)\s+grid=\(([^)]*)\)\s+(.*)', l)
    if m: rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(6).strip()))
starts=[i for i,r in enumerate(rows) if 'time_embed' in r[4]]
step=rows[starts[2]:starts[3]]
t0=step[0][1]
print("--- step 2: every launch (queue, start offset us, duration us, gap us, kernel)")
for r in step: print(f"q{r[0]} {r[1]-t0:8.1f} {r[2]:7.1f} {r[3]:7.1f}  {r[4][:60]}")
P
