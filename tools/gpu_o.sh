cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
B="timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing"
for i in 1 2; do
$B --wgrad-side 0 --text-side 1 > $O/bench_w0t1_$i.json 2>> $O/err0
$B --wgrad-side 1 --text-side 1 > $O/bench_w1t1_$i.json 2>> $O/err0
done
$B --wgrad-side 0 --text-side 1 --precision bf16 > $O/bench_w0t1_bf16.json 2>> $O/err0
$B --wgrad-side 1 --text-side 1 --precision bf16 > $O/bench_w1t1_bf16.json 2>> $O/err0
$B --wgrad-side 0 --text-side 0 --precision bf16 > $O/bench_w0t0_bf16.json 2>> $O/err0
for f in $O/bench_*.json; do echo $f $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"loss": [0-9.]*' $f | head -1); done; grep -v amdgpu $O/err0 | tail -3
