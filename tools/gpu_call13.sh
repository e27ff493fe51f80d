cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c13; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "other_baseline" 2>&1 | tail -25 ) > $O/pytest.log 2>&1
bash tools/gpu_pmc.sh c13 > $O/pmc.out 2>&1
