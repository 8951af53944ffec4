cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py tests/test_hip_net.py tests/test_hip_full_size.py tests/test_hip_trajectory.py tests/test_hip_full_trajectory.py tests/test_hip_teacher_forced.py tests/test_hip_uninit.py -x -q 2>&1 | grep -v "^PARITY test_hip" | tail -16
for i in 1 2; do
echo "== B=1 base"; BDM_ATTN_KSPLIT=1 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== C4 base"; BDM_ATTN_KSPLIT=1 python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
echo "== C4 new"; python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
done
