cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for L in libegovlp_hip.so libegovlp_hip_nt519.so; do for h in 2 4; do
  echo "== $L EGV_TIME_HPW=$h"
  EGV_TIME_HPW=$h EGOVLP_HIP_LIB=$PWD/egovlp_amd/$L timeout 200 python tools/attn_time.py 2>&1 | grep "time attention bwd"
done; done > $O/time_bwd.txt 2>&1
cat $O/time_bwd.txt
