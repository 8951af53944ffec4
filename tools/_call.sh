cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_sampler.py tests/test_hip_trajectory.py tests/test_hip_full_trajectory.py tests/test_hip_full_size.py tests/test_hip_cli.py tests/test_rng.py -x -q 2>&1 | tail -14
python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -2
python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -2
