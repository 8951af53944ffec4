cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "base: $(BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/attn_bench.py 2>&1 | tail -1)"
echo "new : $(python tools/attn_bench.py 2>&1 | tail -1)"
done
