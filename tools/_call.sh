cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for c in c3 c4 c5; do python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05/bench_$c.json; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_$c.json').read()); print('$c', round(d['value'],3), d['unit'], round(d['ms_per_step'],1), d['config'].get('workload','')[:80])"; done
bash tools/pmc_mfma.sh > gpurun_out/r05/mfma_busy.txt 2>&1; head -30 gpurun_out/r05/mfma_busy.txt
