cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "egomcq" 2>&1 | grep -v amdgpu | tail -25 ) > $O/pytest_graph.log 2>&1
timeout 300 python tools/eval_graph_bench.py > $O/eval_graph.txt 2>&1
cat $O/pytest_graph.log; grep -v amdgpu $O/eval_graph.txt | tail -5
