#!/bin/bash
# Round-end evidence on a GPU box: bench line, rocprofv3 --kernel-trace --stats of the same command, per-launch view of one
# forward, per-kernel HBM traffic (two --pmc passes).  Outputs under gpurun_out/ (copy the summaries into profiles/).
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r02}; COMMIT=${2:-unknown}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python bench.py --steps 2 --warmup 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null; echo "rocprof rc=$?"
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trf -o fwd -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
python3 $R/tools/trace_summary.py $(find /tmp/trf -name "*kernel_trace.csv" | head -1) 80 > $R/gpurun_out/${TAG}_forward_launches.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p1 -o f -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p2 -o w -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find /tmp/p1 -name "*counter_collection.csv" | head -1) $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_hbm_traffic.csv $R/gpurun_out/${TAG}_pmc_dispatches.csv
BDM_ABI_LOG=$R/gpurun_out/${TAG}_abi_log.json python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
python3 $R/tools/pmc_to_json.py $R/gpurun_out/${TAG}_pmc_dispatches.csv $R/gpurun_out/${TAG}_abi_log.json $COMMIT $R/gpurun_out/${TAG}_pmc_traffic.json
head -3 $R/gpurun_out/${TAG}_bench.json | cut -c1-600
