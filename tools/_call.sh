cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
BDM_RECORD_DURATIONS=gpurun_out/r6/durations.json timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -170 > gpurun_out/r6/r06_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r6/r06_gpu_suite.txt
bash tools/run_round_profile.sh r06 66b42fc > gpurun_out/r6/round_profile.log 2>&1
tail -4 gpurun_out/r6/r06_gpu_suite.txt; tail -3 gpurun_out/r6/round_profile.log | cut -c1-300
