cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_trajectory.py -q -x -k "prior_loop or launch_tape" 2>&1 | tail -15
for i in 1 2; do
  for e in 0 1; do
    BDM_PVD_TAPE=$e python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('PVD tape $e', 'run $i', 'shapes/s', round(d['value'], 4), 'ms_per_trajectory', round(d['ms_per_step'], 1), 'timed', d['roofline'].get('launches_timed'), 'total', d['roofline'].get('launches_total'))"
  done
done
python -m pytest tests/test_hip_full_trajectory.py tests/test_hip_sampler.py tests/test_hip_cli.py tests/test_hip_full_size.py -q -x 2>&1 | tail -12
