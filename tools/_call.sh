# scratch script of the builder's gpurun calls
cd $GRAFT_REPO_ROOT
echo "base B=16:"; TB=16 TN=4096 BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/forward_hash.py 2>&1 | tail -2
echo "new B=16:"; TB=16 TN=4096 python tools/forward_hash.py 2>&1 | tail -2
for i in 1 2 3; do
echo "== step base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
done
echo "== B=1 base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
