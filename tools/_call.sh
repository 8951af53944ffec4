cd $GRAFT_REPO_ROOT
echo "base:"; BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/forward_hash.py 2>&1 | tail -2
echo "new:"; python tools/forward_hash.py 2>&1 | tail -2
echo "base B=16:"; TB=16 TN=4096 BDM_LIB_PATH=bdm_amd/libbdm_hip_base.so python tools/forward_hash.py 2>&1 | tail -2
echo "new B=16:"; TB=16 TN=4096 python tools/forward_hash.py 2>&1 | tail -2
