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


def test_gemm_desc_layout_matches_header():
    from egovlp_amd._lib import GemmDesc
    # 3x(ptr,ptr,i64) ... the struct is plain C: check total size = sum with natural alignment
    assert ctypes.sizeof(GemmDesc) == 176
    assert GemmDesc.M.offset == 48 and GemmDesc.bias.offset == 72 and GemmDesc.partial.offset == 168


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
