mkdir -p gpurun_out/r05
python tools/error_budget.py pc2 4096 1 > gpurun_out/r05/error_budget.txt 2>&1
python tools/error_budget.py pvd 4096 1 >> gpurun_out/r05/error_budget.txt 2>&1
python tools/error_budget.py pc2 1024 2 >> gpurun_out/r05/error_budget.txt 2>&1
cat gpurun_out/r05/error_budget.txt
