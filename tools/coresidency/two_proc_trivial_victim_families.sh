#!/bin/bash
# The trivial gather8 victim (tools/coresidency/two_proc_repro.hip) next to the LIBRARY's kernel families looping in a second (Python) process.
export LD_LIBRARY_PATH=$PWD/bdm_amd:$LD_LIBRARY_PATH
R=tools/bin/two_proc_repro
for k in ${FAMILIES:-conv_h2_32 conv_h2_16 conv_h2_8 attention to_h2 pw gn fps ball_query sparse forward}; do
  echo "--- aggressor (second process, tools/coresidency/aggressor.py): $k"
  timeout 120 python tools/coresidency/aggressor.py $k 45 > /dev/null 2>&1 &
  pid=$!
  sleep 14
  timeout 120 $R 100 gather8 2>&1 | grep "gather8\|first differing"
  wait $pid
done
