cd $GRAFT_REPO_ROOT
O=gpurun_out/c18; mkdir -p $O
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --precision bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('planes', d['ms_per_step'])" >> $O/ab.log
EGV_LN_DY_F32=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-kernel-timing --precision bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32   ', d['ms_per_step'])" >> $O/ab.log
done
