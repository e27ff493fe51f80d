cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
( timeout 300 python tools/step_phases.py 20 2>&1 | grep -v amdgpu ) > $O/step_phases.txt
cat $O/step_phases.txt
( EGV_WGRAD_SIDE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 --text-side 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one stream:', d['value'], d['ms_per_step'])" )
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg --wgrad-side 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('text side only:', d['value'], d['ms_per_step'])" )
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fast-mode --no-trajectory --no-h2d-leg --no-dp-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default:', d['value'], d['ms_per_step'])" )
