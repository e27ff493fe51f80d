# same-box interleaved A/B of environment settings: bash tools/gpu_ab_env2.sh <tag> <reps> "ENV=.. ENV2=.." "ENV=.." ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; reps=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
F="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --no-grad-err"
for rep in $(seq 1 $reps); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout 600 python bench.py $F > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$O/ab_${i}_$rep.json")); print("rep $rep [$E]", d["value"], d["ms_per_step"], d["config"]["precision"])
except Exception as e:
    print("rep $rep [$E] FAILED", e)
PY
  done
done 2>&1 | tee $O/ab.txt
