cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for i in 1 2 3; do
echo "== step base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
done
echo "== B=1 base"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
BDM_RECORD_DURATIONS=gpurun_out/r05/durations.json python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/r05/suite_3.txt
tail -64 gpurun_out/r05/suite_3.txt
