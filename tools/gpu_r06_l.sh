cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "text_attention or divided_attention" 2>&1 | grep -v "amdgpu\|^$" | tail -30 ) > $O/pytest_attn.txt 2>&1
tail -12 $O/pytest_attn.txt | cut -c1-300
