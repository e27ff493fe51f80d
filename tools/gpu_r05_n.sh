cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; shift
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2 3; do
  for L in "$@"; do
    F=egovlp_amd/libegovlp_hip.so; [ $L != main ] && F=egovlp_amd/libegovlp_hip_$L.so
    echo -n "lib=$L rep=$rep " >> $O/ab.txt
    ( EGOVLP_HIP_LIB=$GRAFT_REPO_ROOT/$F timeout 300 $B 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" ) >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
