cd $GRAFT_REPO_ROOT
BDM_LIB_PATH=bdm_amd/libbdm_hip_diltiming.so python tools/vox_dilate_probe.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_hip_dense.py tests/test_hip_compact_tail.py tests/test_hip_net.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo "== step base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
done
