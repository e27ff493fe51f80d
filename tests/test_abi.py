"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/egovlp_hip.h declares, and the ctypes prototypes cover exactly that set (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "egovlp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(egv_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for s in ("egv_gemm_nt", "egv_layernorm_fwd", "egv_layernorm_bwd", "egv_divided_attn_fwd", "egv_divided_attn_bwd",
              "egv_text_attn_fwd", "egv_text_attn_bwd", "egv_egonce_fwd_bwd", "egv_adamw_multi", "egv_patch_gather"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from egovlp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libegovlp_hip.so not built (run __graft_entry__.build())")
    h = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(h, s), s


def test_ctypes_prototypes_match_header_symbol_set():
    from egovlp_amd import _lib
    assert sorted(_lib.PROTOTYPES.keys()) == declared_symbols()


ABI_STRUCTS = {"egv_gemm_desc": "GemmDesc", "egv_block_geom": "BlockGeom", "egv_block_params": "BlockParams",
               "egv_block_bwd_io": "BlockBwdIO", "egv_text_geom": "TextGeom", "egv_text_params": "TextParams"}


def test_gemm_desc_pinned_offsets():
    """The descriptor every GEMM call passes: size and the offsets the documentation quotes."""
    from egovlp_amd._lib import GemmDesc
    assert ctypes.sizeof(GemmDesc) == 208
    assert GemmDesc.M.offset == 48 and GemmDesc.bias.offset == 72 and GemmDesc.partial.offset == 168
    assert GemmDesc.trans.offset == 176 and GemmDesc.aux_bf16.offset == 180 and GemmDesc.colsum.offset == 184 and GemmDesc.grid_cap.offset == 192
    assert GemmDesc.out_fmt.offset == 196 and GemmDesc.out_bf.offset == 200


def test_every_abi_struct_layout_matches_header(tmp_path):
    """Compile the header with the host C compiler and compare sizeof / offsetof of EVERY field of ALL SIX structs of the C ABI
    (egv_gemm_desc, egv_block_geom / _params / _bwd_io, egv_text_geom / _params) with the ctypes mirrors the Python host side
    passes; the struct declarations in the header and the mirrors must also name the same fields in the same order."""
    import shutil
    import subprocess
    from egovlp_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no host C compiler")
    hdr = open(os.path.join(ROOT, "include", "egovlp_hip.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "egovlp_hip.h"', 'int main(void){']
    for cname, pyname in ABI_STRUCTS.items():
        mirror = getattr(_lib, pyname)
        fields = [f[0] for f in mirror._fields_]
        # field names as the header declares them, in order
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr_nc, flags=re.S).group(1)
        declared = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first = True
            for part in stmt.split(","):
                toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part.split("[")[0])
                declared.append(toks[-1])
                first = False
        assert declared == fields, (cname, declared, fields)
        lines.append('printf("%%zu\\n", sizeof(%s));' % cname)
        lines += ['printf("%%zu\\n", offsetof(%s, %s));' % (cname, f) for f in fields]
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    i = 0
    for cname, pyname in ABI_STRUCTS.items():
        mirror = getattr(_lib, pyname)
        fields = [f[0] for f in mirror._fields_]
        assert out[i] == ctypes.sizeof(mirror), (cname, out[i], ctypes.sizeof(mirror))
        assert out[i + 1:i + 1 + len(fields)] == [getattr(mirror, f).offset for f in fields], cname
        i += 1 + len(fields)


def test_product_raises_without_gpu_tensors():
    """No CPU fallback: ops refuse host tensors instead of silently computing on the CPU."""
    import torch
    from egovlp_amd import _lib, ops
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    with pytest.raises(_lib.EgovlpHipError):
        ops.split_f32(torch.zeros(4, 8), 3)


def test_missing_library_fails_loudly(monkeypatch):
    from egovlp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libegovlp_hip.so")
    with pytest.raises(_lib.EgovlpHipError):
        _lib.lib()


def test_small_gemm_splitk_policy():
    """Host-side kernel choice for NT GEMMs (egovlp_amd.ops): token-major shapes go to the big-tile kernel un-split, the
    DistilBERT-sized ones (M = 1024) get 2-4 k-slices of at least 6 k-steps each, nothing is split beyond 4."""
    from egovlp_amd import ops
    for (M, N, K) in [(25120, 768, 768), (25120, 2304, 768), (25120, 768, 3072), (50192, 768, 768), (16400, 1024, 4096)]:
        assert ops.uses_big_gemm(M, N, K) and ops.auto_ksplit_nt(M, N, K) == 1
    want = {(1024, 768, 768): 4, (1024, 2304, 768): 2, (1024, 3072, 768): 1, (1024, 768, 3072): 4}
    for shape, ks in want.items():
        assert not ops.uses_big_gemm(*shape)
        assert ops.auto_ksplit_nt(*shape) == ks
    for M in (32, 128, 512, 1024, 3140):
        for N in (256, 768, 2304, 3072):
            for K in (256, 768, 3072):
                ks = ops.auto_ksplit_nt(M, N, K)
                assert 1 <= ks <= 4 and (ks == 1 or (K // 32) // ks >= 6 or (K // 32) // 6 >= ks)


def test_abi_version_handshake():
    """A binding written against another header must fail at LOAD time (round-5 advisor: the ABI grew -- a trailing struct field, new
    passes / mode codes -- with nothing a stale caller could trip over): egv_abi_check compares the caller's EGV_ABI_VERSION and struct
    sizes with the library's; egovlp_amd._lib.lib() refuses a mismatch."""
    from egovlp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libegovlp_hip.so not built (run __graft_entry__.build())")
    h = ctypes.CDLL(_lib.LIB_PATH)
    h.egv_abi_check.argtypes = [ctypes.c_int32] + [ctypes.c_int64] * 4
    h.egv_abi_check.restype = ctypes.c_int32
    sizes = [ctypes.sizeof(c) for c in (_lib.GemmDesc, _lib.BlockGeom, _lib.BlockParams, _lib.BlockBwdIO)]
    txt = open(os.path.join(ROOT, "include", "egovlp_hip.h")).read()
    assert int(re.search(r"#define EGV_ABI_VERSION (\d+)", txt).group(1)) == _lib.ABI_VERSION == h.egv_version()
    assert h.egv_abi_check(_lib.ABI_VERSION, *sizes) == 0
    assert h.egv_abi_check(_lib.ABI_VERSION - 1, *sizes) == 1                   # an older binding
    assert h.egv_abi_check(_lib.ABI_VERSION, sizes[0] - 8, *sizes[1:]) == 1     # a caller whose egv_gemm_desc lacks the last field
    assert h.egv_abi_check(_lib.ABI_VERSION, sizes[0], sizes[1] - 4, *sizes[2:]) == 1
    saved, saved_ver = _lib._lib, _lib.ABI_VERSION
    try:
        _lib._lib, _lib.ABI_VERSION = None, saved_ver + 1
        with pytest.raises(_lib.EgovlpHipError, match="ABI mismatch"):
            _lib.lib()
    finally:
        _lib._lib, _lib.ABI_VERSION = saved, saved_ver
