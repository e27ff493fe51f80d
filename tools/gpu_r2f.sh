# round-2 call F: HW bf16 conversion in the GEMM epilogue (A/B vs the pre-F3 library), stagger probe, full op tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_ops.log 2>&1
for rep in 1 2; do
  for lib in libegovlp_hip.so libegovlp_hip_seg.so; do
    EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/$lib timeout 300 python tools/gemm_bench.py 3 1 2>&1 | grep -v amdgpu > $O/gemm_real_mixed_${lib}_$rep.txt
  done
done
timeout 300 python tools/stagger_probe.py 3 2>&1 | grep -v amdgpu > $O/stagger_probe.txt
timeout 300 python tools/stagger_probe.py 1 2>&1 | grep -v amdgpu >> $O/stagger_probe.txt
cat $O/pytest_ops.log; head -8 $O/gemm_real_mixed_libegovlp_hip.so_1.txt; tail -n 1 $O/gemm_real_*; cat $O/stagger_probe.txt
