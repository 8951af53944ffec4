#!/bin/bash
# Round 2 of the aggressor bisect: tools/coresidency/two_proc_aggressors.sh named the sparse first convolution as the only kernel family of a
# second process that disturbs the victim.  Here: its three kernels one by one, and trivial division kernels, all torch-free.
# usage: tools/coresidency/two_proc_aggressors2.sh [reps] [seconds]
REPS=${1:-120}; SECS=${2:-30}
victim() { tools/bin/two_proc_repro $REPS - 2>&1 | grep -v "^ *first differing" | awk '{print "      " $0}' | cut -c1-170; }
for kind in features gemm_s3 gather f64div f32div; do
  echo "=== aggressor: two_proc_repro --aggress $kind"
  tools/bin/two_proc_repro --aggress $kind $SECS > /tmp/aggr.log 2>&1 &
  pid=$!
  sleep 2
  victim
  wait $pid; tail -1 /tmp/aggr.log
done
