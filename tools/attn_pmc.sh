#!/bin/bash
# Counter picture of the flash-attention kernel alone (tools/attn_bench.py): wave-cycle buckets, LDS conflicts, MFMA busy.
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/ap$1; timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d /tmp/ap$1 -o a -- python3 $R/tools/attn_bench.py > /dev/null 2>&1; }
run 1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
run 2 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run 3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"
python3 - <<'PY'
import csv, glob, collections
for i in (1, 2, 3):
    f = glob.glob(f"/tmp/ap{i}/**/*counter_collection.csv", recursive=True)
    if not f: print("pass", i, "no file"); continue
    agg = collections.defaultdict(float); n = collections.Counter(); dur = 0.0; seen = set()
    for r in csv.DictReader(open(f[0])):
        if "attn_flash" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = max(len(seen), 1)
    print(f"pass {i}: {k} dispatches, {dur / k:.1f} us each; per dispatch:")
    for c, v in sorted(agg.items()): print(f"    {c:34s} {v / k:16.0f}")
PY
