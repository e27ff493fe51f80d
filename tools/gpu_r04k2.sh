cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "divided_attention or attention" 2>&1 | grep -v "amdgpu\|^$" | tail -5 ) > $O/t_attn2.txt 2>&1
tail -2 $O/t_attn2.txt
( ATT_B=16 ATT_T=16 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu ) > $O/attn_time_T16_v2.txt 2>&1
grep -H time $O/attn_time_T16_v2.txt
