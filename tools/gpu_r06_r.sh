# persistent space-attention forward: correctness (every attention test) + isolated timing A/B
O=gpurun_out/r06r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16bwd.py tests/test_gpu_f16x2.py -m gpu -x -q -k "attention or attn" 2>&1 | tail -5 > $O/pytest_attention.txt
for p in 0 1 0 1; do echo "== EGV_ATTN_PERSIST=$p"; EGV_ATTN_PERSIST=$p timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu.ids; done > $O/attn_time_ab.txt 2>&1
cat $O/pytest_attention.txt; cat $O/attn_time_ab.txt
