cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_net.py tests/test_hip_dense.py tests/test_hip_full_size.py -q -x 2>&1 | tail -30
