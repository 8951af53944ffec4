# scratch script of the builder's gpurun calls (the last one: the GPU suite twice, for flakiness)
cd $GRAFT_REPO_ROOT
for i in 1 2; do python -m pytest tests -m gpu -q 2>&1 | tail -2; done
