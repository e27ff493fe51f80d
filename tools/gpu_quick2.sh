# usage: bash tools/gpu_quick2.sh <tag> : ops tests + one bench line (about 1 GPU-minute)
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2 ) > $O/pytest_ops.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 > $O/bench.txt
cat $O/pytest_ops.log $O/bench.txt
