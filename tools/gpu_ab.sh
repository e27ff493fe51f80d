# same-box A/B of gemm_big variants built as separate libraries: bash tools/gpu_ab.sh <tag> <lib suffixes...>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/egovlp_amd/libegovlp_hip$v.so
  for shape in "25120 768 3072" "25120 2304 768" "25120 768 768"; do
    EGOVLP_HIP_LIB=$lib timeout 100 python tools/gemm_trace.py $shape 2>&1 | grep -E "span" | sed "s/^/[$v] /"
  done
done
done | tee $O/ab.log
