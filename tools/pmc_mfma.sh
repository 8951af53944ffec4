#!/bin/bash
# MFMA pipe utilisation per kernel of one PC2 forward: SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CU_CYCLES (one --pmc pass).
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d /tmp/m1 -o a -- python3 $R/tools/trace_forward.py pc2 > /tmp/m1.log 2>&1
echo "rc=$?"; tail -3 /tmp/m1.log
cp $(find /tmp/m1 -name "*counter_collection.csv" | head -1) $R/gpurun_out/mfma_1.csv 2>/dev/null; ls -la $R/gpurun_out/mfma_1.csv
python3 - <<PY
import csv, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); cnt = collections.Counter(); seen=set()
for r in csv.DictReader(open("$R/gpurun_out/mfma_1.csv")):
    key = (re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", ""), r["Grid_Size"])
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if (key, r["Dispatch_Id"]) not in seen:
        seen.add((key, r["Dispatch_Id"])); cnt[key]+=1; dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':46s} {'grid':>9s} {'n':>3s} {'us':>7s} | MFMA busy / CU busy   MOPS f32   MOPS f16 per launch")
for key in sorted(agg, key=lambda k: -dur[k])[:40]:
    v = agg[key]; n = cnt[key]
    b = v.get("SQ_BUSY_CU_CYCLES", 0) or 1
    print(f"{key[0][:46]:46s} {key[1]:>9s} {n:3d} {dur[key]/n:7.1f} | {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/b:8.3f}   {v.get('SQ_INSTS_VALU_MFMA_MOPS_F32',0)/n:12.0f} {v.get('SQ_INSTS_VALU_MFMA_MOPS_F16',0)/n:12.0f}")
PY
