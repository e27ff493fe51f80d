cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04u; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | grep "f16x2 \|passed\|failed\|Error\|assert" | cut -c1-300 ) > $O/t_f16x2.txt 2>&1
cat $O/t_f16x2.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 ) > $O/smoke.log 2>&1; cat $O/smoke.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --frames 16 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg > $O/bench_config4.json 2>> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg > $O/bench_config5.json 2>> $O/bench_default.err
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['config']['precision'], d.get('step_mfma_frac'), (d.get('roofline') or {}).get('frac'), d.get('grad_rel_err',{}).get('max'), {k:v for k,v in (d.get('trajectory') or {}).items() if 'gap' in k or 'drift_f' in k or 'drift_m' in k})"; done
