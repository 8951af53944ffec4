#!/bin/bash
# Two copies of tools/coresidency/two_proc_race.py at the same time on ONE GPU, under a matrix of switches; prints both verdicts per row.
# usage: tools/coresidency/two_proc_matrix.sh [reps]   (run from the repo root on the GPU box)
REPS=${1:-24}
run_pair() {  # label, env assignments...
  local label=$1; shift
  echo "=== $label"
  (env "$@" timeout 600 python tools/coresidency/two_proc_race.py $REPS > /tmp/race_b.log 2>&1 &)
  env "$@" timeout 600 python tools/coresidency/two_proc_race.py $REPS 2>&1 | grep -v "^$" | cut -c1-400
  sleep 2; echo "--- second process"; grep -v "^$" /tmp/race_b.log | cut -c1-400
}
run_single() {
  local label=$1; shift
  echo "=== $label (ONE process)"
  env "$@" timeout 600 python tools/coresidency/two_proc_race.py $REPS 2>&1 | grep -v "^$" | cut -c1-400
}
run_single "baseline" X=1
run_pair "baseline" X=1
run_pair "sampler chain inline (BDM_SIDE_STREAM=0)" BDM_SIDE_STREAM=0
run_pair "all inline (BDM_SIDE_STREAM=0 RACE_POINT_STREAM=0 RACE_SIDE_PLAN=0)" BDM_SIDE_STREAM=0 RACE_POINT_STREAM=0 RACE_SIDE_PLAN=0
run_pair "voxel plans on main only (RACE_SIDE_PLAN=0)" RACE_SIDE_PLAN=0
run_pair "device sync after every operator (RACE_SYNC=1)" RACE_SYNC=1
run_pair "no caching allocator (PYTORCH_NO_CUDA_MEMORY_CACHING=1)" PYTORCH_NO_CUDA_MEMORY_CACHING=1
run_pair "PC2 denoiser" RACE_MODEL=pc2
