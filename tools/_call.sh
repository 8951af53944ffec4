cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/run_round_profile.sh r05 de2e1d2
BDM_RECORD_DURATIONS=gpurun_out/r05/durations.json python -m pytest tests -m gpu -q 2>&1 | tail -150 > gpurun_out/r05/gpu_suite.txt
BDM_LIB_PATH=bdm_amd/libbdm_hip_experimental.so python -m pytest tests/test_hip_small_glue.py -q 2>&1 | tail -3 > gpurun_out/r05/gpu_suite_experimental.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05/smoke.txt 2>&1
for c in c3 c4 c5; do python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05/bench_$c.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_$c.json').read()); print('$c', round(d['value'],3), d['unit'], round(d['ms_per_step'],1))"; done
for i in 1 2; do
  for v in r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'run $i', 'shapes/s', round(d['value'], 4), 'ms_per_trajectory', round(d['ms_per_step'], 1), 'conv8 us', round(d['roofline'].get('avg_launch_us', 0), 1))"
  done
done > $R/gpurun_out/r05/round_over_round.txt
cd $R
for i in 1 2 3; do
  for v in r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=16 N=4096"; python tools/replay_host_time.py 16 4096 2>&1 | grep replayed | tail -1
  done
done >> $R/gpurun_out/r05/round_over_round.txt
for v in r4 r5 r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=1 N=1024"; python tools/replay_host_time.py 1 1024 2>&1 | grep replayed | tail -1
done >> $R/gpurun_out/r05/round_over_round.txt
for v in r4 r5 r4 r5; do
    if [ $v = r4 ]; then cd $R/gpurun_tmp/r4; else cd $R; fi
    echo "== $v B=8 N=8192"; python tools/replay_host_time.py 8 8192 2>&1 | grep replayed | tail -1
done >> $R/gpurun_out/r05/round_over_round.txt
cd $R
bash tools/pmc_mfma.sh > gpurun_out/r05/mfma_busy.txt 2>&1
tail -3 gpurun_out/r05/gpu_suite.txt; tail -2 gpurun_out/r05/smoke.txt; cat gpurun_out/r05/round_over_round.txt
