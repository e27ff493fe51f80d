# round 6, session d: stage A of the fp16 backward at the benchmark size -- same box, interleaved A/B of the backward precision
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
F="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in 1 2; do
  for B in bf16 f16; do
    EGV_X2_BWD=$B timeout 600 python bench.py $F > $O/bench_bwd_${B}_$rep.json 2> $O/bench_bwd_${B}_$rep.err
    python - <<PY
import json
d=json.load(open("$O/bench_bwd_${B}_$rep.json"))
print("$B rep $rep", d["value"], d["ms_per_step"], d["config"]["precision"], d.get("grad_rel_err",{}).get("max"), d.get("grad_rel_err",{}).get("per_tensor"), d.get("hbm_reserved_gb"))
PY
  done
done 2>&1 | tee $O/ab_bwd_precision.txt
