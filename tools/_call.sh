cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_small_glue.py -x -q 2>&1 | tail -3
BDM_TAIL_SMALL_GENERIC=1 python -m pytest tests/test_hip_small_glue.py -x -q -k "bit_identical" 2>&1 | tail -2
for n in 64 256; do python tools/tail_bench.py $n 2>&1 | grep -A9 "head=True"; done
echo "== glue on"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed
echo "== BDM_SMALL_GLUE=0"; BDM_SMALL_GLUE=0 python tools/replay_host_time.py 16 4096 2>&1 | grep replayed
