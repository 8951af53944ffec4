cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
BDM_RECORD_DURATIONS=gpurun_out/r6/durations.json timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -170 > gpurun_out/r6/r06_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r6/r06_gpu_suite.txt
echo "== EXPERIMENTAL=1 build: the half-tile list convolution's cases (tests/test_hip_dense.py, tests/test_hip_compact_tail.py)" >> gpurun_out/r6/r06_gpu_suite.txt
BDM_LIB_PATH=$GRAFT_REPO_ROOT/bdm_amd/libbdm_hip_experimental.so timeout 900 python -m pytest tests/test_hip_dense.py tests/test_hip_compact_tail.py -m gpu -q -k "dilated or output_stationary or compact or second_conv or se_gate or voxel_lists" 2>&1 | grep -v PARITY | tail -4 >> gpurun_out/r6/r06_gpu_suite.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/r06_bench_driver_form.json 2> gpurun_out/r6/r06_bench_driver_form.err
BDM_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6/r06_bench_gpus2_shared.json 2> gpurun_out/r6/r06_bench_gpus2_shared.err
tail -12 gpurun_out/r6/r06_gpu_suite.txt | cut -c1-200; python tools/bench_digest.py gpurun_out/r6/r06_bench_driver_form.json | head -3; python tools/bench_digest.py gpurun_out/r6/r06_bench_gpus2_shared.json | head -2
