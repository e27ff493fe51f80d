# same-box A/B of bench.py under different environments / flags: bash tools/gpu_ab_env.sh <tag> <reps> "<env and flags 1>" "<env and flags 2>" ...
# each spec: "VAR=val VAR2=val -- --flag x" (the part before -- is the environment, after it extra bench.py flags)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; REPS=$2; shift; shift
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg"
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    E="${spec%%--*}"; F="${spec#*--}"; [ "$F" == "$spec" ] && F=""
    echo -n "[$spec] rep=$rep " >> $O/ab.txt
    ( env $E timeout 300 $B $F 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" ) >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
