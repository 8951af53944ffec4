# Replayed-step timeline under a few settings: where does the main queue idle?  usage (GPU box, repo root): bash tools/trace_step_variants.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/trs; rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o st -- python3 $R/tools/trace_step.py > /dev/null 2>&1
  F=$(find /tmp/trs -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_timeline.py $F > /tmp/tl.txt; python3 $R/tools/trace_step_summary.py /tmp/tl.txt | sed -n 3,5p | cut -c1-260
}
echo "== default"; run
for q in 1 2 8; do echo "== GPU_MAX_HW_QUEUES=$q"; export GPU_MAX_HW_QUEUES=$q; run; unset GPU_MAX_HW_QUEUES; done
echo "== SIDE_PLAN=0"; export TRACE_SIDE_PLAN=0; run; unset TRACE_SIDE_PLAN
