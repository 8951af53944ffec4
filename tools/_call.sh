cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_net.py tests/test_hip_trajectory.py -q -x 2>&1 | grep -v "^PARITY" | tail -5
for i in 1 2 3; do
echo "== step base"; BDM_DECODER_PLAN=0 python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
done
echo "== B=1 base"; BDM_DECODER_PLAN=0 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== B=1 base"; BDM_DECODER_PLAN=0 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
