cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2 ) > $O/pytest_ops.log 2>&1
for rep in 1 2; do
for L in libegovlp_hip_nt0.so libegovlp_hip.so libegovlp_hip_nt23.so libegovlp_hip_nt103.so libegovlp_hip_nt135.so libegovlp_hip_nt263.so libegovlp_hip_nt519.so libegovlp_hip_nt1015.so; do
  v=$(EGOVLP_HIP_LIB=$PWD/egovlp_amd/$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$L rep$rep $v"
done; done > $O/nt.txt 2>&1
cat $O/pytest_ops.log $O/nt.txt
