cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "side_stream" 2>&1 | grep -E "side streams|passed|failed|Error" | tail -15 ) > $O/pytest_side.log 2>&1
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing"
$B --wgrad-side 0 --text-side 0 > $O/bench_w0t0.json 2> $O/err0
$B --wgrad-side 1 --text-side 0 > $O/bench_w1t0.json 2>> $O/err0
$B --wgrad-side 0 --text-side 1 > $O/bench_w0t1.json 2>> $O/err0
$B --wgrad-side 1 --text-side 1 > $O/bench_w1t1.json 2>> $O/err0
$B --wgrad-side 0 --text-side 0 > $O/bench_w0t0_again.json 2>> $O/err0
$B --wgrad-side 1 --text-side 1 > $O/bench_w1t1_again.json 2>> $O/err0
cat $O/pytest_side.log; for f in $O/bench_*.json; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done; grep -v amdgpu $O/err0 | tail -3
