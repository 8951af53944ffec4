cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
echo "base B=16:"; TB=16 TN=4096 BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/forward_hash.py 2>&1 | tail -2
echo "new B=16:"; TB=16 TN=4096 python tools/forward_hash.py 2>&1 | tail -2
python -m pytest tests/test_hip_compact_tail.py tests/test_hip_net.py -x -q 2>&1 | grep -v "^PARITY" | tail -2
cd /tmp && export TMPDIR=/tmp
for v in base new base new; do
  if [ $v = base ]; then export BDM_LIB_PATH=$R/bdm_amd/libbdm_hip_base.so; else unset BDM_LIB_PATH; fi
  rm -rf /tmp/trf_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trf_$v -o fwd -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
  echo "== $v"
  python3 $R/tools/trace_summary.py $(find /tmp/trf_$v -name "*kernel_trace.csv" | head -1) 120 | grep "to_h2_rows\|se_rows_partial\|devox_rows\|class_constants"
done
