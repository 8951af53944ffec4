cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for cfg in "8 8192" "8 4096" "4 4096" "2 4096"; do echo "== B N = $cfg"; python tools/sparse_os_probe.py $cfg list 2>&1 | grep -v amdgpu; done > gpurun_out/r6/c13_small_batch.txt 2>&1
cat gpurun_out/r6/c13_small_batch.txt
