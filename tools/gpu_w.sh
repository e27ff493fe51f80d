cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "adamw_overlapped" 2>&1 | grep -v amdgpu | tail -15 ) > $O/pytest_adamw.log 2>&1
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing"
for i in 1 2; do
$B --adamw-overlap 0 > $O/bench_a0_$i.json 2>> $O/err0
$B --adamw-overlap 1 > $O/bench_a1_$i.json 2>> $O/err0
done
$B --adamw-overlap 1 --wgrad-side 1 > $O/bench_a1w1.json 2>> $O/err0
cat $O/pytest_adamw.log; for f in $O/bench_*.json; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done; grep -v amdgpu $O/err0 | tail -3
