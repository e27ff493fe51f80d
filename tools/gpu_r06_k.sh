cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "precision_guard" 2>&1 | grep -v "amdgpu\|^$" | tail -30 ) > $O/pytest_guard.txt 2>&1
tail -30 $O/pytest_guard.txt | cut -c1-600
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d.get('precision_guard'))"
