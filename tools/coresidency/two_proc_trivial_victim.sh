#!/bin/bash
# The trivial gather8 victim of tools/coresidency/two_proc_repro.hip next to every aggressor kind, aggressor in a SECOND PROCESS (and alone).
export LD_LIBRARY_PATH=$PWD/bdm_amd:$LD_LIBRARY_PATH
R=tools/bin/two_proc_repro
echo "--- alone"; timeout 120 $R 100 gather8 2>&1 | grep "gather8\|static inputs"
for k in ${KINDS:-copy lds32k lds128k f32div f64div exit8k exit48k noexit48k exit48k_mfma noexit48k_mfma features gather pw gemm_s3_all gemm_s3}; do
  echo "--- aggressor (second process): $k"
  timeout 100 $R --aggress $k 45 > /dev/null 2>&1 &
  pid=$!
  sleep 6
  timeout 120 $R 100 gather8 2>&1 | grep "gather8\|first differing"
  wait $pid
done
