cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_dense.py -x -q -k "attention" 2>&1 | tail -4
