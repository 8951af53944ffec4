cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r6/r06_bench_$c.json 2> gpurun_out/r6/r06_bench_$c.err; python tools/bench_digest.py gpurun_out/r6/r06_bench_$c.json | head -2; done
for f in 0 1 0 1; do echo "NO_WAIT=$f $(TRACE_NO_WAIT=$f python tools/replay_host_time.py 16 4096 2>&1 | tail -1)"; done > gpurun_out/r6/c16_nowait.txt 2>&1
cat gpurun_out/r6/c16_nowait.txt
timeout 1500 python -m pytest tests -m gpu_slow -q -s 2>&1 | tail -40 > gpurun_out/r6/r06_gpu_slow.txt; tail -5 gpurun_out/r6/r06_gpu_slow.txt
