# scratch script of the builder's gpurun calls
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py tests/test_hip_net.py tests/test_hip_full_size.py -x -q 2>&1 | grep -v "^PARITY" | tail -3
