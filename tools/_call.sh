# scratch script of the builder's gpurun calls: `gpurun -- 'bash tools/_call.sh'` (rewritten per call; the calls of a round are listed in
# profiles/r05_latency_chain_ab.txt and profiles/README.md).  As committed: the GPU suite + the smoke check.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
