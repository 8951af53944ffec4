#!/bin/bash
# Round 4: what does a kernel that runs next to sparse_gemm_s3_kernel lose -- its barrier, its LDS contents or its registers?
REPS=${1:-200}; SECS=${2:-20}
echo "=== probes alone"; tools/bin/two_proc_repro $REPS probe 2>&1 | grep "^probe"
for kind in gemm_s3 exit48k_mfma features; do
  echo "=== probes next to --inproc $kind (same process, second stream)"
  tools/bin/two_proc_repro --inproc $kind $SECS $REPS probe 2>&1 | grep "^probe\|aggressor"
done
echo "=== probes next to a second PROCESS running gemm_s3"
tools/bin/two_proc_repro --aggress gemm_s3 $SECS > /tmp/aggr.log 2>&1 &
pid=$!
sleep 2
tools/bin/two_proc_repro $REPS probe 2>&1 | grep "^probe"
wait $pid
