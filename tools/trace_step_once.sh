cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/trs; timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o st -- python3 $R/tools/trace_step.py > /dev/null 2>&1
F=$(find /tmp/trs -name "*kernel_trace.csv" | head -1); python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/tl_now.txt; python3 $R/tools/trace_step_summary.py $R/gpurun_out/tl_now.txt | sed -n 3,5p | cut -c1-330
