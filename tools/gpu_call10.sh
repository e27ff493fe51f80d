cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c10; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -20 ) > $O/pytest.log 2>&1
( EGV_TIME_HPW=4 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -5 ) > $O/pytest4.log 2>&1
bash tools/gpu_prof.sh c10 bf16
EGV_TIME_HPW=4 bash tools/gpu_prof.sh c10b bf16
