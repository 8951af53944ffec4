# scratch script of the builder's gpurun calls: `gpurun -- 'bash tools/_call.sh'` (rewritten per call; the calls of a round are listed in
# profiles/README.md).  As committed: the GPU suite + the smoke check.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
