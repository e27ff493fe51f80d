# round 5, GPU call C: single-product proj (attention writes fp16(value) as its second plane) -- tests, policy A/B, configs 4 / 5.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_block.py -q -s -x 2>&1 | grep -v "amdgpu\|^$" | tail -70 ) > $O/pytest_new.txt 2>&1
tail -3 $O/pytest_new.txt
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" 2>&1 | grep -v "amdgpu\|^$" | tail -5 ) > $O/pytest_attn.txt 2>&1
tail -2 $O/pytest_attn.txt
( timeout 600 python -m pytest tests/test_gpu_model.py -q -s -x -k "golden_in_the_f16x2_mode" 2>&1 | grep -v "amdgpu\|^$" | grep "full B=4\|passed\|failed" ) > $O/pytest_golden.txt 2>&1
cat $O/pytest_golden.txt
bash tools/gpu_ab_env.sh $1 2 "EGOVLP_PRECISION=f16mix EGV_F16_SINGLE=fc2:3,fc1:3,qkv:3" "EGOVLP_PRECISION=f16mix" "EGOVLP_PRECISION=f16mix EGV_F16_SINGLE=fc2:3,fc1:3,qkv:3,proj:4" "EGOVLP_PRECISION=f16mix EGV_F16_SINGLE=fc2:3,fc1:3,qkv:3,proj:0" > /dev/null 2>&1
cat $O/ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 --frames 16 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config4_T16_B16.json 2> $O/bench_c4.err
timeout 600 python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-trajectory --no-h2d-leg --no-dp-leg > $O/bench_config5_vitl14_B16.json 2> $O/bench_c5.err
for f in $O/bench_config4_T16_B16.json $O/bench_config5_vitl14_B16.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print(d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['host_enqueue_ms_from_idle_streams'], 'frac', d['step_mfma_frac'], d['roofline']['frac'], 'grad', d['grad_rel_err']['max'], 'reserved', d['hbm_reserved_gb'], d['alloc_retries'])"; done
