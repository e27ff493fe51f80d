cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "side_stream or golden" 2>&1 | tail -5 ) > $O/pytest_side.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --wgrad-side 0 > $O/bench_side0.json 2> $O/bench0.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --wgrad-side 1 > $O/bench_side1.json 2> $O/bench1.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --wgrad-side 1 --precision bf16 > $O/bench_side1_bf16.json 2> $O/bench1b.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --wgrad-side 0 --precision bf16 > $O/bench_side0_bf16.json 2> $O/bench0b.err
cat $O/pytest_side.log; for f in $O/bench_side*.json; do echo $f; cut -c1-160 $f; done; tail -3 $O/bench1.err
