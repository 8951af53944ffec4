cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_net.py tests/test_hip_full_size.py tests/test_hip_compact_tail.py tests/test_hip_full_trajectory.py tests/test_hip_trajectory.py -x -q 2>&1 | grep -v "^PARITY test_hip" | tail -14
for i in 1 2; do
echo "== base-like (BDM_COMPACT_TAIL=always at 16 too is not the same; reference: previous default via env)"; 
echo "== new default"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
done
