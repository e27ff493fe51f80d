cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -2 ) > $O/pytest_attn.log 2>&1
for rep in 1 2 3; do
  v=$(EGV_TIME_HPW=2 EGOVLP_HIP_LIB=$PWD/egovlp_amd/libegovlp_hip_nt7.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "mask7_hpw2 rep$rep $v"
  v=$(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "mask519_hpw4 rep$rep $v"
done > $O/nt.txt 2>&1
cat $O/pytest_attn.log $O/nt.txt
