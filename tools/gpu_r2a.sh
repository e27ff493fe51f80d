# round-2 call A: GEMM tests on the new epilogue + same-box A/B of the GEMM microbench (new lib vs round-1 lib)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -15 ) > $O/pytest_gemm.log 2>&1
for rep in 1 2; do
  for lib in libegovlp_hip.so libegovlp_hip_r1.so; do
    EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/$lib timeout 300 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_real_mixed_${lib}_$rep.txt
    EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/$lib timeout 300 python tools/gemm_bench.py 1 1 2>&1 | grep -v amdgpu > $O/gemm_real_bf16_${lib}_$rep.txt
  done
done
cat $O/pytest_gemm.log; tail -n 20 $O/gemm_real_mixed_libegovlp_hip.so_1.txt $O/gemm_real_mixed_libegovlp_hip_r1.so_1.txt; tail -n 1 $O/gemm_real_*
