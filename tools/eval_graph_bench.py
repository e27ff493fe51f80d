"""EgoMCQ-shaped forward (1 text query + 5 candidate clips of 4 frames, trainer/trainer_egoclip.py:197-214): eager launches vs
HIP-graph replay (egovlp_amd/graph.py).  usage: python tools/eval_graph_bench.py [clips] [frames]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from egovlp_amd import ops
from egovlp_amd.graph import GraphedForward

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ops.Precision.set("bf16x3")
model = bench.build_model("base_patch16_224", 16).cuda().eval()
g = torch.Generator().manual_seed(1)
data = {"video": torch.randn(clips, T, 3, 224, 224, generator=g).cuda(),
        "text": {"input_ids": torch.randint(1000, 30000, (1, 16), generator=g).cuda(), "attention_mask": torch.ones(1, 16, dtype=torch.long).cuda()}}


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    te, out_e = timeit(lambda: model(data))
    out_e = [o.clone() for o in out_e]
    fwd = GraphedForward(model)
    tg, out_g = timeit(lambda: fwd(data))
print(f"EgoMCQ question ({clips} clips x {T} frames + 1 query): eager {te:.2f} ms, HIP graph replay {tg:.2f} ms ({te / tg:.2f}x), "
      f"captures {fwd.stats['captures']}, max |diff| {max(float((a - b).abs().max()) for a, b in zip(out_e, out_g)):.1e}")
