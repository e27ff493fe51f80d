cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/c17; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
bash tools/gpu_prof.sh c17 bf16
bash tools/gpu_prof.sh c17 mixed
