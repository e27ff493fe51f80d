cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04t; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -40 ) > $O/t_f16x2.txt 2>&1
tail -25 $O/t_f16x2.txt | cut -c1-220
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "f16x2 and not B16 and not vitl and not large" 2>&1 | grep -v "amdgpu\|^$" | tail -20 ) > $O/t_model_f16x2.txt 2>&1
tail -8 $O/t_model_f16x2.txt | cut -c1-220
for rep in 1 2; do for pr in mixed f16x2; do
  timeout 600 python bench.py --precision $pr --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_${pr}_$rep.json 2> $O/bench_${pr}_$rep.err
done; done
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['loss'], d.get('grad_rel_err',{}).get('max'))"; done
