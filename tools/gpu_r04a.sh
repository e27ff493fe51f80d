# round 4, first contact of the f16f6 path with hardware: kernel tests, model tests in the new mode, same-box bench A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16f6.py -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -150 ) > $O/t_f16f6.txt 2>&1
( timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "f16f6 or full_size_train_step" 2>&1 | grep -v "amdgpu\|^$" | tail -150 ) > $O/t_model.txt 2>&1
for rep in 1 2; do
for prec in mixed f16f6; do
  ( timeout 600 python bench.py --precision $prec --steps 20 --warmup 5 --no-trajectory --no-h2d-leg --no-fast-mode --no-cpu-baseline > $O/bench_${prec}_$rep.json 2> $O/bench_${prec}_$rep.err )
done; done
grep -h "passed\|failed\|error" $O/t_f16f6.txt $O/t_model.txt | tail; for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('host_enqueue_ms_per_step'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
