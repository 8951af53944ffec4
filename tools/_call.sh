cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_net.py -m gpu -q -x -k "conv3d or golden or forced" 2>&1 | grep -v PARITY | tail -6 > gpurun_out/r6/c11_tests.txt
for pp in 0 1 0 1; do echo "== PINGPONG=$pp"; BDM_CONV_PINGPONG=$pp BENCH_S3=0 python tools/conv_h2_bench.py 16 2>&1 | grep -v amdgpu; done > gpurun_out/r6/c11_conv.txt 2>&1
for pp in 0 1 0 1; do echo "PINGPONG=$pp $(BDM_CONV_PINGPONG=$pp python tools/replay_host_time.py 16 4096 2>&1 | tail -1)"; done > gpurun_out/r6/c11_step.txt 2>&1
cat gpurun_out/r6/c11_tests.txt gpurun_out/r6/c11_conv.txt gpurun_out/r6/c11_step.txt
