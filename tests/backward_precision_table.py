"""What would an fp16 backward buy?  ("table first".)   TEST INFRASTRUCTURE ONLY.

    python tests/backward_precision_table.py [--config b4|vitl] [--threads N]

The CPU oracle runs one EgoNCE training step under autograd; for every Linear of the video tower the upstream gradient dY, the saved
input X and the weight W are captured, and the two GEMMs of its backward -- dW = dY^T X (weight gradient) and dX = dY W (data
gradient) -- are recomputed in fp64 from operands rounded (a) to bf16 (what the single-pass backward of the benchmarked mode
multiplies) and (b) to fp16 after scaling dY by a power of two S (an fp16 backward with a static loss scale).  Reported per op kind
(worst and median over the blocks): the magnitude range of dY, the fraction of its elements that fp16 would flush or denormalise at
S, and the rel-L2 error of dW / dX under both roundings.  These are PER-GEMM rounding errors; through the depth of the tower they
compound (the measured end-to-end weight-gradient error of the bf16 backward is 2e-3 .. 2.5e-2, tests/test_gpu_model.py)."""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import egovlp_oracle as O                               # noqa: E402
from precision_table import build                                   # noqa: E402


class _Tap:
    def __init__(self, names):
        self.names, self.rec = names, []

    def __getattr__(self, k):
        return getattr(F, k)

    def linear(self, x, w, b=None):
        y = F.linear(x, w, b)
        name = self.names.get(id(w))
        if name is not None and ".blocks." in name and y.requires_grad:
            ent = {"name": name, "x": x.detach().reshape(-1, x.shape[-1]), "w": w.detach()}
            y.register_hook(lambda g, e=ent: e.__setitem__("dy", g.detach().reshape(-1, g.shape[-1])))
            self.rec.append(ent)
        return y


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="b4")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    sd, batch, vcfg = build(a.config)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tap = _Tap({id(v): k for k, v in sd.items()})
    saved = O.F
    O.F = tap
    try:
        te, ve = O.frozen_in_time(batch, sd, vcfg, O.TextCfg())
        loss, _ = O.egoclip_loss(te, ve, batch["noun_vec"], batch["verb_vec"])
        loss.backward()
    finally:
        O.F = saved
    B = batch["video"].shape[0]
    print(f"# config {a.config}: B={B} depth={vcfg.depth} D={vcfg.embed_dim}, loss {float(loss):.4f}; dY = d loss / d (Linear output), loss = mean over the batch")
    print("# fp16: normal range 6.1e-5 .. 65504, subnormals down to 6e-8; 'below normal' loses relative precision, 'flushed' (< 3e-8) becomes 0")
    bf = lambda t: t.to(torch.bfloat16).double()
    kinds = {}
    for e in tap.rec:
        if "dy" not in e:
            continue
        k = e["name"].split(".blocks.")[1].split(".", 1)[1].rsplit(".", 1)[0]       # timeattn.qkv, attn.proj, mlp.fc1, ...
        kinds.setdefault(k, []).append(e)
    amax_all = max(float(e["dy"].abs().max()) for v in kinds.values() for e in v)
    S_safe = 2.0 ** math.floor(math.log2(32768.0 / amax_all))                       # largest power of two keeping max |S dY| < 2^15
    print(f"# max |dY| over all Linears {amax_all:.3e}  ->  largest safe static scale S = 2^{int(math.log2(S_safe))}")
    print(f"{'op':14s} {'max |dY|':>10s} {'median |dY|':>12s} | fp16 at S=1: {'below normal':>12s} {'flushed':>9s} | at S=2^{int(math.log2(S_safe))}: {'below normal':>12s} {'flushed':>9s} |"
          f" {'dW bf16':>9s} {'dW fp16':>9s} {'dX bf16':>9s} {'dX fp16':>9s}   (rel-L2, worst block / median block)")
    for k, ents in kinds.items():
        rows = []
        for e in ents:
            dy, x, w = e["dy"].double(), e["x"].double(), e["w"].double()
            ady = dy.abs()
            nz = ady[ady > 0]
            st = {"max": float(ady.max()), "med": float(nz.median())}
            for S, tag in ((1.0, "1"), (S_safe, "s")):
                st["sub" + tag] = float((nz * S < 6.1035e-5).double().mean())
                st["fl" + tag] = float((nz * S < 2.98e-8).double().mean())
            dW, dX = dy.t() @ x, dy @ w
            h16 = lambda t, S=1.0: (t * S).float().clamp(-65504, 65504).to(torch.float16).double() / S
            st["dW_bf"] = rel(bf(e["dy"]).t() @ bf(e["x"]), dW)
            st["dW_h"] = rel(h16(e["dy"], S_safe).t() @ h16(e["x"]), dW)
            st["dX_bf"] = rel(bf(e["dy"]) @ bf(e["w"]), dX)
            st["dX_h"] = rel(h16(e["dy"], S_safe) @ h16(e["w"]), dX)
            rows.append(st)
        worst = lambda key: max(r[key] for r in rows)
        med = lambda key: sorted(r[key] for r in rows)[len(rows) // 2]
        print(f"{k:14s} {worst('max'):10.2e} {med('med'):12.2e} | {'':12s} {worst('sub1'):12.3f} {worst('fl1'):9.4f} | {'':10s} {worst('subs'):12.4f} {worst('fls'):9.5f} |"
              f" {worst('dW_bf'):.1e}/{med('dW_bf'):.1e} {worst('dW_h'):.1e}/{med('dW_h'):.1e} {worst('dX_bf'):.1e}/{med('dX_bf'):.1e} {worst('dX_h'):.1e}/{med('dX_h'):.1e}", flush=True)


if __name__ == "__main__":
    main()
