# in-kernel split-K reduce of the weight gradients: parity + same-box A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm_tn or splitk" 2>&1 | grep -v "amdgpu\|^$" | tail -15 ) > $O/t_tn.txt 2>&1
for rep in 1 2 3; do for e in 1 0; do
  EGV_SPLITK_IN_KERNEL=$e timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_inkernel${e}_$rep.json 2> $O/bench_inkernel${e}_$rep.err
done; done
tail -3 $O/t_tn.txt
for f in $O/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; done
