cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "== step default (tail where a head follows)"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== step tail_all"; BDM_SMALL_GLUE=tail_all python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
done
echo "== B=1 default"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=1 tail_all"; BDM_SMALL_GLUE=tail_all python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
