# usage: bash tools/gpu_cfg45.sh <tag> : BASELINE configs 4 (T=16, B=16) and 5 (ViT-L/14, B=16) at full size, one bench line each
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 600 python bench.py --frames 16 --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_config4_T16_B16.json 2> $O/err4
timeout 600 python bench.py --arch large_patch14_224 --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_config5_vitl14_B16.json 2> $O/err5
for f in $O/bench_config*.json; do echo $f; cut -c1-330 $f; grep -o '"fast_mode_bf16": {[^}]*}' $f; done; tail -2 $O/err4 $O/err5
