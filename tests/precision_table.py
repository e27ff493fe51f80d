"""Forward-precision table on the CPU oracle.  TEST INFRASTRUCTURE ONLY.

    python tests/precision_table.py [--config b4|t16|vitl] [--schemes ...] [--per-op]

For every scheme of tests/quant_emul.py the oracle's forward is re-run with all matrix products computed from rounded operands,
and the embeddings / EgoNCE loss are compared with the fp32 oracle (the bar: 1e-3 relative, BASELINE.json north_star).  `--per-op`
additionally maps which op kinds tolerate ONE bf16 product when everything else runs bf16x3.  Output is committed under profiles/.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd.synth import synth_state_dict, synth_batch          # noqa: E402
from oracle import egovlp_oracle as O                               # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from quant_emul import Policy, QuantisedOracle                      # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def build(config):
    from egovlp_amd.model.schema import state_dict_schema
    if config == "b4":
        vcfg, T, B, kw = O.VideoCfg(), 4, 4, {}
    elif config == "t16":
        vcfg, T, B, kw = O.VideoCfg(num_frames=16), 16, 2, {"num_frames": 16}
    elif config == "vitl":
        vcfg = O.VideoCfg(patch_size=14, embed_dim=1024, depth=24, num_heads=16, num_frames=4)
        T, B, kw = 4, 2, {"embed_dim": 1024, "depth": 24, "patch_size": 14, "num_frames": 4}
    else:
        raise ValueError(config)
    schema = state_dict_schema(**kw)
    sd = synth_state_dict(schema, seed=0)
    batch = synth_batch(B, T=T, L=32, seed=1234, ragged=True)
    return sd, batch, vcfg


def run(sd, batch, vcfg, policy):
    with torch.no_grad():
        if policy is None:
            te, ve = O.frozen_in_time(batch, sd, vcfg, O.TextCfg())
        else:
            with QuantisedOracle(O, sd, policy):
                te, ve = O.frozen_in_time(batch, sd, vcfg, O.TextCfg())
        loss, _ = O.egoclip_loss(te, ve, batch["noun_vec"], batch["verb_vec"])
    return te, ve, float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="b4")
    ap.add_argument("--schemes", nargs="*", default=["bf16", "fp16", "bf16x3", "bf16+e4m3", "bf16+e5m2", "bf16+e2m3", "bf16+e3m2",
                                                     "bf16+e2m1", "bf16+e2m1/e4m3", "bf16+e2m3/e4m3"])
    ap.add_argument("--scaling", default="ceil")
    ap.add_argument("--per-op", action="store_true")
    ap.add_argument("--per-op-fp16", action="store_true",
                    help="ONE fp16 product for one op kind (and unions) on top of the f16x2 mode (VERDICT r04 next-1a)")
    ap.add_argument("--per-block-fp16", action="store_true",
                    help="fc2 in ONE fp16 product in every block, fc1 in ONE fp16 product in a subset of the blocks")
    ap.add_argument("--sensitivity", action="store_true", help="ONE op kind (qkv / fc1 / fc2) in ONE block in one fp16 product, everything else f16x2")
    ap.add_argument("--policy", nargs="*", default=[], metavar="K_FC2,K_FC1,K_QKV",
                    help="the op runs ONE fp16 product from that block on, two before (egovlp_amd.ops.single_product_policy: depth / 4 each)")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    sd, batch, vcfg = build(a.config)
    t0 = time.time()
    rt, rv, rl = run(sd, batch, vcfg, None)
    print(f"# config {a.config}: B={batch['video'].shape[0]} T={batch['video'].shape[1]} depth={vcfg.depth} D={vcfg.embed_dim}; "
          f"fp32 oracle {time.time() - t0:.1f}s, loss {rl:.6f}; MX scale choice: {a.scaling}")
    print(f"{'scheme':34s} {'text emb':>10s} {'video emb':>10s} {'|d loss|':>10s}   s")
    for s in a.schemes:
        t0 = time.time()
        te, ve, l = run(sd, batch, vcfg, Policy(default=s, scaling=a.scaling))
        print(f"{s:34s} {rel(te, rt):10.2e} {rel(ve, rv):10.2e} {abs(l - rl):10.2e}   {time.time() - t0:.0f}", flush=True)
    if a.per_op:
        print("# one op kind at a time in ONE bf16 product, everything else bf16x3")
        for kind in ("patch", "qkv", "qk", "pv", "proj", "fc1", "fc2", "text_lin", "text_qk", "text_pv", "head"):
            te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **{kind: "bf16"}))
            print(f"{'bf16x3, ' + kind + '=bf16':34s} {rel(te, rt):10.2e} {rel(ve, rv):10.2e} {abs(l - rl):10.2e}", flush=True)
    if a.per_op_fp16:
        base = dict(qkv="fp16x2:6", fc1="fp16x2:6", fc2="fp16x2:6")
        print("# base = the f16x2 mode (video blocks' qkv / fc1 / fc2 in two fp16 products, bf16x3 elsewhere); listed kinds in ONE fp16 product")
        rows = [(k,) for k in ("patch", "qkv", "qk", "pv", "proj", "fc1", "fc2", "text_lin", "text_qk", "text_pv", "head")]
        rows += [("fc1", "fc2"), ("fc1", "fc2", "pv"), ("fc1", "fc2", "proj"), ("fc1", "fc2", "pv", "proj"),
                 ("fc1", "fc2", "pv", "proj", "qk"), ("fc1", "fc2", "pv", "proj", "qk", "qkv"),
                 ("text_lin", "text_qk", "text_pv"), ("fc2", "text_lin", "text_qk", "text_pv"),
                 ("fc1", "fc2", "text_lin", "text_qk", "text_pv")]
        for kinds in rows:
            pol = dict(base)
            pol.update({k: "fp16" for k in kinds})
            t0 = time.time()
            te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **pol))
            print(f"{'f16x2, fp16 x1: ' + ' + '.join(kinds):58s} {rel(te, rt):10.2e} {rel(ve, rv):10.2e} {abs(l - rl):10.2e}   {time.time() - t0:.0f}",
                  flush=True)
    if a.per_block_fp16:
        base = dict(qkv="fp16x2:6", fc1="fp16x2:6", fc2="fp16")
        dep = vcfg.depth
        print("# base = f16x2 mode with fc2 in ONE fp16 product in every block; fc1 in ONE fp16 product in the listed blocks only")
        sets = [("none", []), ("last quarter", range(dep - dep // 4, dep)), ("last half", range(dep // 2, dep)),
                ("first half", range(0, dep // 2)), ("last two thirds", range(dep // 3, dep)), ("all", range(dep))]
        for label, blocks in sets:
            pol = dict(base)
            pol.update({f"fc1@{b}": "fp16" for b in blocks})
            t0 = time.time()
            te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **pol))
            print(f"{'fc2 x1 everywhere, fc1 x1 in: ' + label:58s} {rel(te, rt):10.2e} {rel(ve, rv):10.2e} {abs(l - rl):10.2e}   {time.time() - t0:.0f}",
                  flush=True)
    x2 = dict(qkv="fp16x2:6", fc1="fp16x2:6", fc2="fp16x2:6")
    if a.sensitivity:
        te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **x2))
        print(f"# {a.config}: base f16x2 video {rel(ve, rv):.3e}", flush=True)
        for op in ("qkv", "fc1", "fc2"):
            for b in range(0, vcfg.depth, 1 if vcfg.depth <= 12 else 2):
                pol = dict(x2)
                pol[f"{op}@{b}"] = "fp16"
                te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **pol))
                print(f"{op}@{b:<3d} video {rel(ve, rv):.3e} dloss {abs(l - rl):.2e}", flush=True)
    for spec in a.policy:
        k2, k1, kq = [int(x) for x in spec.split(",")]
        pol = dict(x2)
        for op, k in (("fc2", k2), ("fc1", k1), ("qkv", kq)):
            for b in range(k, vcfg.depth):
                pol[f"{op}@{b}"] = "fp16"
        te, ve, l = run(sd, batch, vcfg, Policy(default="bf16x3", **pol))
        print(f"{a.config}: one fp16 product from block fc2 >= {k2}, fc1 >= {k1}, qkv >= {kq} (two products before; bf16x3 elsewhere):  "
              f"text {rel(te, rt):.2e} video {rel(ve, rv):.2e} dloss {abs(l - rl):.2e}", flush=True)


if __name__ == "__main__":
    main()
