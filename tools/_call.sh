cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
timeout 600 python tools/sparse_os_probe.py 16 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r6/c05_probe8.txt
cat gpurun_out/r6/c05_probe8.txt
