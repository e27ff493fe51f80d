cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04n; mkdir -p $O
for rep in 1 2; do for l in "" _tmfall; do
  EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench${l}_$rep.json 2> $O/bench${l}_$rep.err
done; done
for f in $O/bench*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'host' in k})"; done
