cd $GRAFT_REPO_ROOT
python tools/tape_torch_ops.py 16 4096 2>&1 | grep -v amdgpu.ids | tail -8
python -m pytest tests/test_hip_net.py tests/test_hip_trajectory.py tests/test_hip_uninit.py tests/test_hip_sampler.py tests/test_hip_ops.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
echo "== step legacy copies"; BDM_LEGACY_COPIES=1 python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
echo "== step new"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
done
for i in 1 2; do
echo "== B=1 legacy"; BDM_LEGACY_COPIES=1 python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
echo "== B=1 new"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
done
