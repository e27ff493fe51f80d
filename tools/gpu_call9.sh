cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c9; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -20 ) > $O/pytest.log 2>&1
bash tools/gpu_prof.sh c9 bf16
