"""Micro-benchmark of egv_gemm_nt on the hot-path shapes (uniform random operands, HIP-event timed).
usage: [BENCH_EPI=real|f32|bf16] [BENCH_SINGLE=0..7] [EGOVLP_HIP_LIB=...] python tools/gemm_bench.py [passes [bwd_passes]]
BENCH_SINGLE (passes = 2 only): ops.F16_SINGLE_BITS of the Linears that run ONE fp16 product (1 fc1, 2 fc2, 4 qkv) -- the later blocks of 'f16mix'.
BENCH_EPI=real (default): every shape runs with the epilogue and output formats the EgoClip step gives it in the mode
"forward = passes, backward = bwd_passes" (planes for qkv / h / dZ, fp32 + residual for proj / fc2, GELU + saved pre-activation
for fc1, GELU' for the fc2 dgrad, split-K slabs + column sums for the wgrads); f32 / bf16: plain fp32 / bf16-plane output + bias
(the round-1 table).  The wgrad rows use the TN kernel.  Prints one line per shape and the aggregate."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egovlp_amd import ops  # noqa: E402

_MAIN = __name__ == "__main__"
passes_list = [int(sys.argv[1])] if _MAIN and len(sys.argv) > 1 else [3, 1]
bwd_passes = int(sys.argv[2]) if _MAIN and len(sys.argv) > 2 else 1
EPI = os.environ.get("BENCH_EPI", "real")
SINGLE = int(os.environ.get("BENCH_SINGLE", "0"))
M = int(os.environ.get("BENCH_TOKENS", 32 * 785))
D, H3, HD = 768, 2304, 3072
dev = "cuda"


def rnd(r, c, p, role=0, plain16=False):
    x = torch.rand(r, c, device=dev) * 2 - 1
    if plain16:              # ONE plane of plain fp16 (fmt 'f16'): every operand of the fp16 backward (dY, X, W^T)
        return ops.Planes(x.to(torch.float16), None, r, c, "f16")
    if p == 2:               # f16x2 operands: role 0 = first operand (activations), 1 = second (weights)
        return ops.f16x2_encode(x, role)
    if p == 4:               # ONE fp16 product: a plain fp16 activation plane x the weight's f16x2 encoding (plane 1 = fp16(W))
        return ops.f16x2_encode(x, 1) if role else ops.Planes(x.to(torch.float16), None, r, c, "f16")
    return ops.split_f32(x, p)[0]


def cases(P, Pb, single=0):
    """(name, flops, callable) in the order of one SpaceTimeBlock forward + backward."""
    single = single if P == 2 else 0
    P_qkv, P_fc1, P_fc2 = (4 if single & 4 else P), (4 if single & 1 else P), (4 if single & 2 else P)
    out = []

    h16 = Pb == 4           # the fp16 backward (round 6): fp16 dY / X / W^T planes, fp16 saved gelu', fp16-split qkv planes, no bf16 copies

    def nt(name, m, n, k, p, kw=None, bwd=False):
        a, b = rnd(m, k, p, 0, plain16=bwd and h16), rnd(n, k, p, 1, plain16=bwd and h16)
        bias = torch.zeros(n, device=dev)
        if EPI != "real":
            if EPI == "bf16":
                o = ops.empty_planes(m, n, 1, dev)
                out.append((name, m, n, k, p, lambda: ops.gemm_nt(a, b, passes=p, out_planes=o, bias=bias)))
            else:
                o = torch.empty(m, n, device=dev)
                out.append((name, m, n, k, p, lambda: ops.gemm_nt(a, b, passes=p, out_f32=o, bias=bias)))
            return
        out.append((name, m, n, k, p, lambda: ops.gemm_nt(a, b, passes=p, **kw(bias))))

    def tn(name, m, n, k, p):
        a, b = rnd(k, m, p, plain16=h16 and p == 4), rnd(k, n, p, plain16=h16 and p == 4)
        o = torch.empty(m, n, device=dev)
        out.append((name, m, n, k, p, lambda: ops.gemm_tn(a, b, passes=p, out_f32=o, want_colsum=True, alpha=ops.A1_INV if (h16 and p == 4) else 1.0)))

    res = torch.rand(M, D, device=dev)
    o32 = torch.empty(M, D, device=dev)
    Pa = 3 if P == 2 else P          # the f16x2 mode (P = 2): qkv / fc1 / fc2 forward as two fp16 products, proj / text split-bf16 x3
    P_proj = (4 if single & 8 else 2) if h16 else Pa      # fp16 backward: the proj runs two fp16 products (f16x2 attention output) or one
    qkv_pl = ops.empty_planes_f16x2(M, H3, dev, split=True) if h16 else ops.empty_planes(M, H3, Pa, dev)
    h_pl = ops.empty_planes_f16x2(M, HD, dev, want_bf=not h16, single=bool(single & 2)) if P == 2 else ops.empty_planes(M, HD, P, dev)
    z = torch.empty(M, HD, device=dev, dtype=torch.float16 if h16 else (torch.bfloat16 if Pb == 1 else torch.float32))
    zin = (torch.rand(M, HD, device=dev) * 4 - 2).to(z.dtype)
    dz_pl = ops.empty_planes_f16x2(M, HD, dev, single=True) if h16 else ops.empty_planes(M, HD, Pb, dev)
    dx_pl = ops.empty_planes_f16x2(M, D, dev, single=True) if h16 else ops.empty_planes(M, D, Pb, dev)
    nt("qkv   fwd", M, H3, D, P_qkv, kw=lambda bias: dict(bias=bias, out_planes=qkv_pl))
    nt("proj  fwd", M, D, D, P_proj, kw=lambda bias: dict(bias=bias, residual=res, out_f32=o32))
    nt("fc1   fwd", M, HD, D, P_fc1, kw=lambda bias: dict(bias=bias, act=ops.ACT_GELU, aux_out=z, out_planes=h_pl, aux_is_grad=Pb in (1, 4)))
    nt("fc2   fwd", M, D, HD, P_fc2, kw=lambda bias: dict(bias=bias, residual=res, out_f32=o32))
    nt("fc2 dgrad", M, HD, D, Pb, kw=lambda bias: dict(act=ops.ACT_GELU_BWD, aux_in=zin, out_planes=dz_pl, aux_is_grad=Pb in (1, 4)), bwd=True)
    nt("fc1 dgrad", M, D, HD, Pb, kw=lambda bias: dict(out_f32=o32), bwd=True)
    nt("proj dgrad", M, D, D, Pb, kw=lambda bias: dict(out_planes=dx_pl, grad_out=h16), bwd=True)
    nt("qkv dgrad", M, D, H3, Pb, kw=lambda bias: dict(out_f32=o32), bwd=True)
    tn("qkv wgrad", H3, D, M, Pb)
    tn("proj wgrad", D, D, M, Pb)
    tn("fc1 wgrad", HD, D, M, Pb)
    tn("fc2 wgrad", D, HD, M, Pb)
    nt("text  lin", 1024, D, D, Pa, kw=lambda bias: dict(bias=bias, out_f32=torch.empty(1024, D, device=dev)))
    tn("text wgrad", D, D, 1024, 3 if h16 else Pb)
    return out


for P in (passes_list if __name__ == "__main__" else []):
    Pb = bwd_passes if EPI == "real" else P
    tot_t = tot_f = 0.0
    for name, m, n, k, p, run in cases(P, Pb, SINGLE):
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tf = 2.0 * m * n * k / us / 1e6
        tot_t += us
        tot_f += 2.0 * m * n * k
        pi = 1 if p == 4 else p
        print(f"passes={p} {name:10s} M={m:6d} N={n:5d} K={k:6d}: {us:8.1f} us  {tf:7.1f} TF algorithmic  ({tf * pi:7.1f} TF MFMA issue)")
    print(f"fwd passes={P} (single-product mask {SINGLE}) bwd passes={Pb} epi={EPI} TOTAL {tot_f / tot_t / 1e6:.1f} TF algorithmic, {tot_t:.0f} us per block-equivalent, "
          f"lib={os.path.basename(os.environ.get('EGOVLP_HIP_LIB', 'libegovlp_hip.so'))}")
