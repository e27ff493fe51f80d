# round-end evidence: bash tools/gpu_final.sh <tag>  -> gpurun_out/<tag>/ (GPU tests, smoke, default bench, kernel stats bf16 + mixed, PMC of gemm_big)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/gpu_prof.sh $1 bf16
bash tools/gpu_prof.sh $1 mixed
bash tools/gpu_pmc.sh $1 > $O/pmc.log 2>&1
timeout 100 python tools/gemm_trace.py 25120 2304 768 2>&1 | grep -v amdgpu > $O/tile_timeline_qkv_fwd.txt
timeout 100 python tools/gemm_trace.py 25120 768 3072 2>&1 | grep -v amdgpu > $O/tile_timeline_fc2_fwd.txt
timeout 200 python tools/gemm_bench.py 2>&1 | grep -v amdgpu > $O/gemm_bench.txt
timeout 100 python tools/mfma_peak.py 2>&1 | grep -v amdgpu > $O/mfma_peak.txt
cat $O/pytest_gpu.log $O/smoke.log $O/bench_default.json
