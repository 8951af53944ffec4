#!/bin/bash
# Second round of the two-process experiments (after tools/coresidency/two_proc_matrix.sh showed that neither the side streams nor the absence
# of synchronisation matter): does the difference need the SAME program in both processes (same kernels on the same virtual
# addresses)?  usage: tools/coresidency/two_proc_matrix2.sh [reps]   (repo root, GPU box)
REPS=${1:-16}
pair() {  # label, "env of A", "env of B"
  echo "=== $1"
  (env $3 timeout 600 python tools/coresidency/two_proc_race.py $REPS > /tmp/race_b.log 2>&1 &)
  env $2 timeout 600 python tools/coresidency/two_proc_race.py $REPS 2>&1 | grep -v "^$" | grep "traced calls\|first differing" | cut -c1-300
  sleep 2; echo "--- second process"; grep "traced calls\|first differing" /tmp/race_b.log | cut -c1-300
}
pair "same program, same sizes (baseline)" "X=1" "X=1"
pair "second process runs another problem size (N=1100: other addresses, other grids)" "X=1" "RACE_N=1100"
pair "second process runs the PC2 denoiser" "X=1" "RACE_MODEL=pc2"
pair "one hardware queue per process (GPU_MAX_HW_QUEUES=1)" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=1"
pair "kernels serialised by the runtime (AMD_SERIALIZE_KERNEL=3)" "AMD_SERIALIZE_KERNEL=3" "AMD_SERIALIZE_KERNEL=3"
echo "=== torch-free repro, different data per process"
(tools/bin/two_proc_repro 200 - 2 > /tmp/repro_b.log 2>&1 &)
tools/bin/two_proc_repro 200 - 1 2>&1 | cut -c1-200
sleep 3; echo "--- second process"; cut -c1-200 /tmp/repro_b.log
