cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/run_round_profile.sh r05 803b624
BDM_RECORD_DURATIONS=gpurun_out/r05/durations.json python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/r05/gpu_suite.txt
BDM_LIB_PATH=bdm_amd/libbdm_hip_experimental.so python -m pytest tests/test_hip_small_glue.py -q 2>&1 | tail -3 > gpurun_out/r05/gpu_suite_experimental.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05/smoke.txt 2>&1
for v in r4 r5 r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=1 N=1024"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
done > $R/gpurun_out/r05/c1_step.txt
cd $R
tail -3 gpurun_out/r05/gpu_suite.txt; tail -2 gpurun_out/r05/smoke.txt; cat gpurun_out/r05/c1_step.txt
