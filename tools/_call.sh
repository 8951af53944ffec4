cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
for w in 0 1; do echo "== WHOLE8=$w"; if [ $w = 1 ]; then export BDM_CONV8_WHOLE=1; else unset BDM_CONV8_WHOLE; fi; BENCH_S3=0 python tools/conv_h2_bench.py 16 2>&1 | grep "r= 8"; done > gpurun_out/r6/c12_conv8.txt 2>&1
cat gpurun_out/r6/c12_conv8.txt
