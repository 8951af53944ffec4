#!/bin/bash
# Victim: the torch-free repro (library kernels through the C ABI, one launch + device sync + bitwise compare per repetition).
# Aggressor: a SECOND process looping one kernel family.  Which aggressor makes the victim's repetitions differ?
# usage: tools/coresidency/two_proc_aggressors.sh [reps] [seconds]   (repo root, GPU box)
REPS=${1:-120}; SECS=${2:-25}
victim() { tools/bin/two_proc_repro $REPS lib 2>&1 | grep -v "^ *first differing" | awk '{print "      " $0}' | cut -c1-170; }
echo "=== no aggressor"; victim
for fam in forward conv_h2_32 conv_h2_16 conv_h2_8 to_h2 attention fps ball_query sparse pw gn torch_matmul; do
  echo "=== aggressor: python tools/coresidency/aggressor.py $fam"
  timeout 120 python tools/coresidency/aggressor.py $fam $SECS > /tmp/aggr.log 2>&1 &
  pid=$!
  sleep 9   # import torch + model set-up of the aggressor
  victim
  wait $pid; tail -1 /tmp/aggr.log | cut -c1-120
done
for kind in lds128k lds32k copy; do
  echo "=== aggressor: two_proc_repro --aggress $kind (trivial kernel, no library code)"
  tools/bin/two_proc_repro --aggress $kind $SECS > /tmp/aggr.log 2>&1 &
  pid=$!
  sleep 1
  victim
  wait $pid; tail -1 /tmp/aggr.log
done
