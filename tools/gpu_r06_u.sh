O=gpurun_out/r06u; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_f16bwd.py -m gpu -x -q 2>&1 | tail -3 > $O/pytest_f16bwd.txt; cat $O/pytest_f16bwd.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "whole_batch or trajectory or full_size" 2>&1 | grep -v "^$" > $O/pytest_whole_batch.txt; tail -30 $O/pytest_whole_batch.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
