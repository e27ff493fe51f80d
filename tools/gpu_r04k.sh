# MFMA time attention (4 < T <= 16): parity tests + same-box A/B against the vector-ALU kernels of the previous library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "divided_attention or attention" 2>&1 | grep -v "amdgpu\|^$" | tail -25 ) > $O/t_attn.txt 2>&1
tail -5 $O/t_attn.txt
for l in _attnbefore ""; do
  ( EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so ATT_B=16 ATT_T=16 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu ) > $O/attn_time_T16$l.txt 2>&1
  ( EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so ATT_B=32 ATT_T=8 timeout 300 python tools/attn_time.py 2>&1 | grep -v amdgpu ) > $O/attn_time_T8$l.txt 2>&1
done
grep -H time $O/attn_time_T*.txt
( timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "config4" 2>&1 | grep -v "amdgpu\|^$" | tail -12 ) > $O/t_cfg4.txt 2>&1
tail -4 $O/t_cfg4.txt
for rep in 1 2; do for l in _attnbefore ""; do
  EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip$l.so timeout 600 python bench.py --frames 16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg > $O/bench_cfg4${l}_$rep.json 2> $O/bench_cfg4${l}_$rep.err
done; done
for f in $O/bench_cfg4*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
