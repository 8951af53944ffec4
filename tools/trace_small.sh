#!/bin/bash
# Launch-by-launch timeline of one PC^2 forward for ONE small shape (B=1, N=1024: config C1's step) -> gpurun_out/<tag>_c1_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs
TB=1 TN=1024 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o fwd -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
F=$(find /tmp/trs -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/${TAG}_c1_timeline.txt
python3 $R/tools/trace_summary.py $F 40 > $R/gpurun_out/${TAG}_c1_launches.txt
