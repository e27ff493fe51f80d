"""Run-to-run and path-to-path differences of one SpaceTimeBlock's backward (diagnostic for tests/test_gpu_block.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_block import _block, _run
from egovlp_amd import ops

def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())

for mode in (("bf16x3", "bf16x3"), ("bf16x3", "bf16"), ("bf16", "bf16")):
    for geom in ((8, 4, 196), (2, 16, 196)):
        B, T, n = geom
        blk = _block(768)
        ec = ops.new_context(); ec.set_precision(*mode)
        torch.manual_seed(5)
        x = torch.randn(B, 1 + T * n, 768, device="cuda"); g = torch.randn(B, 1 + T * n, 768, device="cuda") * 0.1
        runs = {}
        for tag, bc in (("c1", True), ("c2", True), ("k1", False), ("k2", False)):
            runs[tag] = _run(blk, ec, x, g, B, T, n, bc, False)
        for a, b in (("c1", "c2"), ("k1", "k2"), ("c1", "k1")):
            d = {"dx": rel(runs[a][1], runs[b][1]), **{k: rel(runs[a][2][k], runs[b][2][k]) for k in runs[a][2]}}
            print(mode, geom, a, "vs", b, {k: "%.1e" % v for k, v in d.items() if v})
