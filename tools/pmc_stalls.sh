#!/bin/bash
# Per-kernel stall picture of one PC2 forward: SQ wave-cycle buckets, cache hit rates, TA / TLB pressure (separate --pmc passes).
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/s1 -o a -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/s2 -o a -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d /tmp/s3 -o a -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/s4 -o a -- python3 $R/tools/trace_forward.py pc2 > /dev/null 2>&1
for i in 1 2 3 4; do cp $(find /tmp/s$i -name "*counter_collection.csv" | head -1) $R/gpurun_out/stalls_$i.csv 2>/dev/null; done
ls -la $R/gpurun_out/stalls_*.csv
