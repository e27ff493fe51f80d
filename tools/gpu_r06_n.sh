cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_block.py -m gpu -q 2>&1 | tail -3 ) | tee $O/pytest_block.txt
bash tools/gpu_ab_env2.sh $1 3 "EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/libegovlp_hip_nomr.so" "EGV_DUMMY=1"
