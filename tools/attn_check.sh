cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python3 $GRAFT_REPO_ROOT/tools/trace_forward.py pc2 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_stats_short.py $(find /tmp/pm -name "*kernel_stats.csv" | head -1) 60 | grep -E "attn_"
cd $GRAFT_REPO_ROOT && timeout 300 python -m pytest tests/test_hip_dense.py -x -q -k "attention" 2>&1 | tail -2
