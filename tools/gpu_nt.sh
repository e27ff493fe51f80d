cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2 3; do
for L in libegovlp_hip_nt515.so libegovlp_hip_nt519.so libegovlp_hip_nt8707.so libegovlp_hip_nt16899.so libegovlp_hip.so; do
  v=$(EGOVLP_HIP_LIB=$PWD/egovlp_amd/$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$L rep$rep $v"
done; done > $O/nt.txt 2>&1
cat $O/nt.txt
