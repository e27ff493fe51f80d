cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O

( timeout 600 python tools/host_overhead.py mixed 2>&1 | grep -v "amdgpu" ) > $O/host_overhead.txt 2>&1
for rep in; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_$rep.json 2> $O/bench_$rep.err
done
tail -4 $O/t_block.txt; head -3 $O/host_overhead.txt
for f in $O/bench_?.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'host' in k})"; done
