# per-kernel HBM traffic of one PC^2 forward: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one)
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p1 -o f -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p2 -o w -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/p1/f_counter_collection.csv /tmp/p2/w_counter_collection.csv $R/gpurun_out/pmc_forward.csv
head -12 $R/gpurun_out/pmc_forward.csv
