cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_small_glue.py tests/test_hip_net.py tests/test_hip_uninit.py tests/test_hip_full_size.py -x -q 2>&1 | tail -14
BDM_LIB_PATH=bdm_amd/libbdm_hip_experimental.so python -m pytest tests/test_hip_small_glue.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo "== step glue off"; BDM_SMALL_GLUE=0 python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== step default (tail+head)"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
done
echo "== B=1 off"; BDM_SMALL_GLUE=0 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=1 default"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=8 N=8192 off"; BDM_SMALL_GLUE=0 python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
echo "== B=8 N=8192 default"; python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
