# round-2 call E: fused three-product GEMM loop -- GEMM tests, A/B vs the three-k-segment library, dropout / dist tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -15 ) > $O/pytest_gemm.log 2>&1
for rep in 1 2; do
  for lib in libegovlp_hip.so libegovlp_hip_seg.so; do
    EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/$lib timeout 300 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_real_mixed_${lib}_$rep.txt
  done
done
( timeout 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_dist.py tests/test_gpu_model.py -m gpu -q -k "dropout or dist or rccl or golden or tiny or train_step" 2>&1 | tail -15 ) > $O/pytest_some.log 2>&1
cat $O/pytest_gemm.log; head -8 $O/gemm_real_mixed_libegovlp_hip.so_1.txt; head -8 $O/gemm_real_mixed_libegovlp_hip_seg.so_1.txt; tail -n 1 $O/gemm_real_*; tail -8 $O/pytest_some.log
