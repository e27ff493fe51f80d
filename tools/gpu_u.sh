cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export EGOVLP_HIP_LIB=$PWD/egovlp_amd/libegovlp_hip_diag.so
for d in 0 1 2; do echo "EGV_TIME_DBG=$d"; EGV_TIME_DBG=$d timeout 200 python tools/attn_time.py 2>&1 | grep "time attention fwd"; done > $O/time_dbg.txt
cat $O/time_dbg.txt
