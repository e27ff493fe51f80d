# C block calls: parity against the per-kernel path + same-box A/B of the host enqueue time and the step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_block.py -m gpu -q -x 2>&1 | grep -v "amdgpu\|^$" | tail -25 ) > $O/t_block.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | grep -v "amdgpu\|^$" | tail -25 ) > $O/t_model.txt 2>&1
for rep in 1 2; do for bc in 1 0; do
  EGV_BLOCK_CALLS=$bc timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_bc${bc}_$rep.json 2> $O/bench_bc${bc}_$rep.err
done; done
tail -4 $O/t_block.txt; tail -4 $O/t_model.txt
for f in $O/bench_bc*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'host' in k})"; done
