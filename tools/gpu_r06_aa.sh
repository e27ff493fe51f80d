O=gpurun_out/r06aa; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_f16bwd.py tests/test_gpu_f16x2.py tests/test_gpu_block.py tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
BENCH_SINGLE=15 timeout 300 python tools/gemm_bench.py 2 4 2>&1 | grep -v amdgpu > $O/gemm_bench_single.txt; cat $O/gemm_bench_single.txt
timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu | tail -8
