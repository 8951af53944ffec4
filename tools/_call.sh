# scratch script of the builder's gpurun calls
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_net.py -x -q 2>&1 | grep -v "^PARITY" | tail -3
