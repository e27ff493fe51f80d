cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( TRACE_DIAG=0x4000 timeout 200 python tools/gemm_trace.py 2304 768 25120 tn 9 2>&1 | grep -v amdgpu ) > $O/tn_trace_qkv_ks9.txt
( TRACE_DIAG=0x4000 timeout 200 python tools/gemm_trace.py 3072 768 25120 tn 7 2>&1 | grep -v amdgpu ) > $O/tn_trace_fc1_ks7.txt
( TRACE_DIAG=0x4000 timeout 200 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -v amdgpu ) > $O/nt_trace_fc1dgrad.txt
cat $O/tn_trace_qkv_ks9.txt | cut -c1-200
cat $O/nt_trace_fc1dgrad.txt | head -12 | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trajectory > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); print(d['value'], d['ms_per_step'], 'dp leg', d['dp_policy_at_world_size_1']['value'], 'h2d', d['with_h2d_uint8']['value'], 'fast', d['fast_mode_bf16']['value'])"
