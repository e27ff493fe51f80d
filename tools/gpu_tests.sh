cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
