# fp16 d_n planes into the LayerNorm backward: tests + same-box A/B against the previous library
O=gpurun_out/r06t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_f16bwd.py tests/test_gpu_block.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest_a.txt
cat $O/pytest_a.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_model.txt
tail -5 $O/pytest_model.txt
bash tools/gpu_ab_env2.sh r06t 3 "EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip_oldln.so" "EGV_X=1"
