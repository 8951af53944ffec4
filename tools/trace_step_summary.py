"""Per-step view of a tools/trace_step.py trace (trace_timeline.py output): for every replayed step, the span from its first to its
last launch, the busy time of the main queue and the idle gaps INSIDE the step (the time between steps under rocprofv3 is the
tracer slowing the host's scheduler step, not the replay).  usage: trace_step_summary.py step_timeline.txt"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'q(\d+) t=\s*([\d.]+)\s+dur=\s*([\d.]+)\s+gap=\s*(-?[\d.]+)\s+grid=\(([^)]*)\)\s+(.*)', l)
    if m:
        rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(6).strip()))
starts = [i for i, r in enumerate(rows) if 'time_embed' in r[4]] + [len(rows)]
for s in range(len(starts) - 1):
    step = rows[starts[s]:starts[s + 1]]
    q0 = [r for r in step if r[0] == 0]
    span = q0[-1][1] + q0[-1][2] - q0[0][1]
    busy = sum(r[2] for r in q0)
    gaps = [(r[3], r[4]) for r in q0[1:] if r[3] > 10]
    print(f"step {s}: {len(step)} launches, main-queue span {span:7.1f} us, busy {busy:7.1f} us, idle inside the step {span - busy:6.1f} us; "
          f"gaps > 10 us: " + ", ".join(f"{g:.0f} us before {n[:28]}" for g, n in gaps))
