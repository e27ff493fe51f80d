cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_trainer.py -m gpu -q 2>&1 | grep -v amdgpu | tail -4 ) > $O/pytest_fix.txt 2>&1
cat $O/pytest_fix.txt
