cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base new base new; do
  if [ $v = base ]; then export BDM_LIB_PATH=$R/bdm_amd/libbdm_hip_base.so; else unset BDM_LIB_PATH; fi
  rm -rf /tmp/trf_$v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trf_$v -o fwd -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
  echo "== $v"
  python3 $R/tools/trace_summary.py $(find /tmp/trf_$v -name "*kernel_trace.csv" | head -1) 80 | grep "conv3d_h2q\|^[0-9]* launches"
done
