#!/bin/bash
# Round 3: sparse_gemm_s3_kernel is the aggressor (tools/coresidency/two_proc_aggressors2.sh).  Which of its properties is it -- reproduced with
# TRIVIAL kernels (48 KB static LDS, workgroups that return at once, matrix cores), and does it need a second process at all?
REPS=${1:-100}; SECS=${2:-25}
victim() { tools/bin/two_proc_repro $REPS "$1" 2>&1 | grep -v "^ *first differing" | awk '{print "      " $0}' | cut -c1-170; }
for kind in exit48k noexit48k exit48k_mfma noexit48k_mfma exit8k gemm_s3; do
  echo "=== second PROCESS: two_proc_repro --aggress $kind"
  tools/bin/two_proc_repro --aggress $kind $SECS > /tmp/aggr.log 2>&1 &
  pid=$!
  sleep 2
  victim lib
  wait $pid; tail -1 /tmp/aggr.log
done
for kind in gemm_s3 exit48k_mfma; do
  echo "=== SAME process, second stream: two_proc_repro --inproc $kind"
  tools/bin/two_proc_repro --inproc $kind $SECS $REPS lib 2>&1 | grep -v "^ *first differing" | awk '{print "      " $0}' | cut -c1-170
done
