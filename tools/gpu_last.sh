cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "tn_wgrad or side_stream or train_step_matches" 2>&1 | tail -2 ) > $O/pytest.log 2>&1
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode 2>/dev/null > $O/bench.json
cat $O/pytest.log; cut -c1-260 $O/bench.json
