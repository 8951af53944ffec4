BDM_WATCHDOG=500 timeout 600 python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; echo rc=$?
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_r01_rocprof.json 2> $R/gpurun_out/bench_r01_rocprof.err; echo rc=$?
mkdir -p $R/gpurun_out/prof_r01d; cp /tmp/prof/*stats*.csv $R/gpurun_out/prof_r01d/
