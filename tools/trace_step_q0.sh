# Replayed-step timelines for the main-queue stall before the first set-abstraction module.  usage (GPU box, repo root): bash tools/trace_step_q0.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/trs; timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o st -- python3 $R/tools/trace_step.py > /dev/null 2>&1
  F=$(find /tmp/trs -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_timeline.py $F > /tmp/tl.txt; python3 $R/tools/trace_step_summary.py /tmp/tl.txt | sed -n 3,5p | cut -c1-330
}
echo "== default"; run; cp /tmp/tl.txt $R/gpurun_out/tl_default.txt
echo "== NO_WAIT (unsafe: main stream never waits)"; export TRACE_NO_WAIT=1; run; unset TRACE_NO_WAIT; cp /tmp/tl.txt $R/gpurun_out/tl_nowait.txt
echo "== DEFER_CHAIN=0"; export TRACE_DEFER_CHAIN=0; run; unset TRACE_DEFER_CHAIN; cp /tmp/tl.txt $R/gpurun_out/tl_nodefer.txt
