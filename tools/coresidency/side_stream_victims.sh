#!/bin/bash
# VERDICT r3 item 5 -- the ONE closing experiment of the co-residency topic: the kernels the product runs on its SIDE streams
# (sampler chain: FPS, ball query, 3-NN search, 32^3 voxel plan; point branch: 1x1 GEMMs) as VICTIMS on the null stream of this
# process while an aggressor loops on a second stream of the SAME process; every victim launch is compared bit by bit with its
# first repetition.  Aggressors: sparse_gemm_s3_kernel (the kernel named in DESIGN.md section 5) and, round 4, the output-
# stationary first convolution that replaced GEMM + gather in the default forward (conv_os).  usage: side_stream_victims.sh [reps]
#   build: hipcc --offload-arch=gfx950 -O2 -I include tools/coresidency/two_proc_repro.hip -o tools/bin/two_proc_repro \
#                -L bdm_amd -l:libbdm_hip.so -Wl,-rpath,'$ORIGIN/../../bdm_amd'
REPS=${1:-1000}
SECS=${2:-300}   # upper bound: the aggressor thread leaves as soon as the victim cases are done
R=tools/bin/two_proc_repro
echo "=== alone ($REPS launches each)"
$R $REPS "sampler" 2>&1 | grep "repetitions\|first differing"
$R $REPS "lib pointwise_conv" 2>&1 | grep "repetitions\|first differing"
for agg in gemm_s3 conv_os; do
  echo "=== aggressor on a second stream of the same process: $agg"
  $R --inproc $agg $SECS $REPS "sampler" 2>&1 | grep "repetitions\|first differing\|aggressor"
  $R --inproc $agg $SECS $REPS "lib pointwise_conv" 2>&1 | grep "repetitions\|first differing\|aggressor"
  echo "--- control: the trivial gather8 victim next to $agg (DESIGN.md section 5: 78 / 200 next to gemm_s3 in round 3)"
  $R --inproc $agg 120 200 "gather8" 2>&1 | grep "repetitions\|aggressor"
done
