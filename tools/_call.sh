cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py tests/test_hip_net.py tests/test_hip_full_size.py tests/test_hip_small_glue.py -x -q 2>&1 | grep -v "^PARITY" | tail -4
for i in 1 2; do echo "== C4 B=8 N=8192"; python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1; done
echo "== B=4 N=4096"; python tools/replay_host_time.py 4 4096 2>&1 | grep replayed | tail -1
