cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "== step base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
done
python -m pytest tests/test_hip_dense.py tests/test_hip_net.py -x -q 2>&1 | grep -v "^PARITY" | tail -3
