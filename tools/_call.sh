cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== B=1 default"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=1 nowait"; TRACE_NO_WAIT=1 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
done
echo "== B=16 nowait"; TRACE_NO_WAIT=1 python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== B=16 default"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
