cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=16 N=4096"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
  done
done
for v in r4 r5 r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=1 N=1024"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
done
