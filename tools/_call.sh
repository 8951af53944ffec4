# scratch script of the builder's gpurun calls (the last one: the live-oracle forms of the long tests, for the record)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu_slow -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r05/gpu_slow.txt
tail -12 gpurun_out/r05/gpu_slow.txt
