cd $GRAFT_REPO_ROOT
for i in 1 2; do
for mb in 512 256 1024 128; do echo "== min_blocks $mb"; BDM_PW_MIN_BLOCKS=$mb python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1; done
for d in 512 2048 4096; do echo "== deep $d"; BDM_PW_DEEP=$d python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1; done
done
