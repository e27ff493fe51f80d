# config 4 (T = 16) time-attention kernels: parity tests + same-box A/B against the library before the change
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "divided_attention" 2>&1 | grep -v "amdgpu\|^$" | tail -15 ) > $O/t_attn.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "config4 and not B16" 2>&1 | grep -v "amdgpu\|^$" | tail -30 ) > $O/t_cfg4.txt 2>&1
for rep in 1 2; do for l in _attnbefore ""; do
  EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so timeout 600 python bench.py --frames 16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_cfg4${l}_$rep.json 2> $O/bench_cfg4${l}_$rep.err
done; done
tail -3 $O/t_attn.txt; tail -3 $O/t_cfg4.txt
for f in $O/bench_cfg4*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
