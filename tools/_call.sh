cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "== C4 base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
echo "== C4 new"; python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
done
for i in 1 2; do
echo "== C2 base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
echo "== C2 new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
done
echo "== B=4 base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 4 4096 2>&1 | grep replayed | tail -1
echo "== B=4 new"; python tools/replay_host_time.py 4 4096 2>&1 | grep replayed | tail -1
