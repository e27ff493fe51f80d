cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for D in 0 90 100; do EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/libegovlp_hip_diag.so EGV_GEMM_DBG=$D timeout 300 python tools/fc1_epilogue_probe.py 2>&1 | grep EGV_GEMM; done; done | tee $O/fc1_epilogue_probe.txt
