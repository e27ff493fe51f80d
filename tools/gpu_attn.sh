cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -6 ) > $O/pytest_attn.log 2>&1
timeout 200 python tools/attn_time.py > $O/attn_time.txt 2>&1
cat $O/pytest_attn.log; grep -v amdgpu $O/attn_time.txt
