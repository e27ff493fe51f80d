# round 5, GPU call E: ViT-L/14 (config 5) -- streaming space-attention forward for the 257-key groups (A/B), then the full GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --arch large_patch14_224 --batch 16 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2; do
  for L in main s18; do
    F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
    echo -n "config5 lib=$L rep=$rep " >> $O/ab_c5.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 400 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_enqueue_ms_from_idle_streams'])" ) >> $O/ab_c5.txt 2>&1
  done
done
cat $O/ab_c5.txt
( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/egovlp_amd/libegovlp_hip_s18.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "attention or attn or vitl" 2>&1 | grep -v "amdgpu\|^$" | tail -3 ) > $O/pytest_s18.txt 2>&1
tail -2 $O/pytest_s18.txt
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu\|^$" | tail -400 ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
