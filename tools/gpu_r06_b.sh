# round 6, session b: bring-up of the fp16 backward
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16bwd.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu\|^$" | tail -80 ) > $O/pytest_f16bwd.txt 2>&1
( timeout 600 python tools/f16bwd_check.py 4 2>&1 | grep -v "amdgpu\|Warning\|detach\|return te\|worst" | tail -40 ) > $O/f16bwd_check.txt 2>&1; ( timeout 600 python tools/f16bwd_check.py 16 2>&1 | grep -v "amdgpu\|Warning\|detach\|return te\|worst" | tail -40 ) > $O/f16bwd_check_b16.txt 2>&1
tail -60 $O/pytest_f16bwd.txt; cat $O/f16bwd_check.txt $O/f16bwd_check_b16.txt
