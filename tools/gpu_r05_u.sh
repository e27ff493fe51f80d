cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" 2>&1 | grep -v "amdgpu\|^$" | tail -8 ) > $O/pytest_attn.txt 2>&1
tail -3 $O/pytest_attn.txt
( timeout 900 python -m pytest tests/test_gpu_model.py -q -s -x -k "other_baseline_configs_full_model and vitl" 2>&1 | grep -v "amdgpu\|^$" | grep "config5\|passed\|failed" | cut -c1-250 ) > $O/pytest_vitl.txt 2>&1
cat $O/pytest_vitl.txt
B="python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2; do
  for L in main bs0; do
    F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
    echo -n "config5 lib=$L rep=$rep " >> $O/ab_c5.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 400 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_enqueue_ms_from_idle_streams'], d['host_wall_ms_between_enqueues'])" ) >> $O/ab_c5.txt 2>&1
  done
done
cat $O/ab_c5.txt
