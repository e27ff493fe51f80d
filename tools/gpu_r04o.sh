cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04o; mkdir -p $O
for lim in 2 0; do
EGV_MAX_STEPS_IN_FLIGHT=$lim timeout 600 python bench.py --frames 16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_cfg4_lim$lim.json 2>> $O/bench.err
EGV_MAX_STEPS_IN_FLIGHT=$lim timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_lim$lim.json 2>> $O/bench.err
done
for f in $O/bench_cfg4_lim*.json $O/bench_lim*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('hbm_reserved_gb'), d.get('alloc_retries'), {k:v for k,v in d.items() if 'host' in k})"; done
