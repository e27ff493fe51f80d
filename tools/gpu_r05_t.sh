# A/B of library variants on the wgrad rows of gemm_bench + step level.   bash tools/gpu_r05_t.sh <tag> <suffix> [<suffix> ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; shift
for L in "$@"; do
  F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
  if [ $L != main ]; then
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or wgrad or tn or linear" 2>&1 | grep -v "amdgpu\|^$" | tail -1 | sed "s/^/lib=$L /" ) >> $O/pytest_variants.txt 2>&1
  fi
  for rep in 1 2; do
    echo "lib=$L rep=$rep" >> $O/gemm_bench_ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 300 python tools/gemm_bench.py 1 1 2>&1 | grep "wgrad" | grep -v text ) >> $O/gemm_bench_ab.txt
  done
done
cat $O/pytest_variants.txt; cat $O/gemm_bench_ab.txt | cut -c1-110
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2 3; do
  for L in "$@"; do
    F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
    echo -n "lib=$L rep=$rep " >> $O/ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 300 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" ) >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
