cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_compact_tail.py -m gpu -x -q -k "dilated or output_stationary or compact or second_conv or se_gate or voxel_lists" 2>&1 | tail -15 > gpurun_out/r6/c02_tests.txt
timeout 600 python tools/sparse_os_probe.py 16 0.5 > gpurun_out/r6/c02_probe.txt 2>&1
timeout 600 python tools/sparse_os_probe.py 16 0.25 >> gpurun_out/r6/c02_probe.txt 2>&1
for f in 0 auto 0 auto; do echo "DIL_TILE=$f $(BDM_DIL_TILE=$f python tools/replay_host_time.py 16 4096 2>&1 | tail -1)"; done > gpurun_out/r6/c02_step.txt 2>&1
cat gpurun_out/r6/c02_tests.txt gpurun_out/r6/c02_probe.txt gpurun_out/r6/c02_step.txt
