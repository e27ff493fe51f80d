"""ctypes binding of libegovlp_hip.so (include/egovlp_hip.h) -- the only native boundary of the package.

There is deliberately NO fallback: if the HIP library is missing or fails to load, every op
raises.  A product path that silently ran on PyTorch eager kernels (or on the CPU oracle)
would void the parity claims.
"""
import ctypes as C
import os

# ORDER MATTERS: PyTorch-ROCm ships its own libamdhip64 (HIP 7.0) while libegovlp_hip.so was linked against
# /opt/rocm's (7.2).  Both have SONAME libamdhip64.so.7, so whichever is loaded first serves the whole process.
# The streams / device pointers we are handed belong to torch's runtime, so torch must be loaded first -- otherwise
# two HIP runtimes coexist and every launch fails with hipErrorNoDevice (100).
import torch  # noqa: F401  (must precede ctypes.CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EGOVLP_HIP_LIB", os.path.join(_HERE, "libegovlp_hip.so"))  # override: diagnostics only

c_p = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
u64 = C.c_uint64


class GemmDesc(C.Structure):
    """egv_gemm_desc (include/egovlp_hip.h)."""
    _fields_ = [
        ("a_hi", c_p), ("a_lo", c_p), ("lda", i64),
        ("b_hi", c_p), ("b_lo", c_p), ("ldb", i64),
        ("M", i32), ("N", i32), ("K", i32), ("passes", i32),
        ("alpha", f32), ("act", i32),
        ("bias", c_p),
        ("residual", c_p), ("ldr", i64),
        ("aux_in", c_p), ("aux_out", c_p), ("ldaux", i64),
        ("out_f32", c_p), ("ldo", i64),
        ("out_hi", c_p), ("out_lo", c_p), ("ldoh", i64),
        ("ksplit", i32), ("accumulate", i32),
        ("partial", c_p),
        ("trans", i32), ("aux_bf16", i32),
        ("colsum", c_p),
        ("grid_cap", i32), ("out_fmt", i32),
        ("out_bf", c_p),
    ]


class BlockGeom(C.Structure):
    """egv_block_geom."""
    _fields_ = [("B", i32), ("T", i32), ("n", i32), ("H", i32), ("D", i32), ("Hd", i32),
                ("fwd_passes", i32), ("bwd_passes", i32), ("train", i32), ("z_bf16", i32), ("eps", f32), ("grid_cap", i32),
                ("f16_single", i32)]


class BlockParams(C.Structure):
    """egv_block_params."""
    _fields_ = [("n3w", c_p), ("n3b", c_p), ("n1w", c_p), ("n1b", c_p), ("n2w", c_p), ("n2b", c_p),
                ("bias", c_p * 6),
                ("w_hi", c_p * 6), ("w_lo", c_p * 6), ("ldw", i64 * 6),
                ("wt_hi", c_p * 6), ("wt_lo", c_p * 6), ("ldwt", i64 * 6)]


class BlockBwdIO(C.Structure):
    """egv_block_bwd_io."""
    _fields_ = [("g_out", c_p), ("g_hi", c_p), ("g_lo", c_p),
                ("x", c_p), ("fwd_arena", c_p), ("bwd_arena", c_p),
                ("d_x", c_p), ("dx_hi", c_p), ("dx_lo", c_p),
                ("grads", c_p),
                ("side_stream", c_p * 6), ("side_event", c_p * 6),
                ("wgrad_ksplit", i32 * 6)]


class TextGeom(C.Structure):
    """egv_text_geom."""
    _fields_ = [("B", i32), ("L", i32), ("H", i32), ("D", i32), ("Hd", i32),
                ("fwd_passes", i32), ("bwd_passes", i32), ("train", i32),
                ("eps", f32), ("attn_p", f32), ("ffn_p", f32), ("grid_cap", i32),
                ("attn_seed", C.c_uint64), ("ffn_seed", C.c_uint64), ("seed_dev", c_p),
                ("nt_ksplit_fwd", i32 * 4), ("nt_ksplit_bwd", i32 * 4), ("wgrad_ksplit", i32 * 4)]


class TextParams(C.Structure):
    """egv_text_params."""
    _fields_ = [("ln1w", c_p), ("ln1b", c_p), ("ln2w", c_p), ("ln2b", c_p),
                ("bias", c_p * 4),
                ("w_hi", c_p * 4), ("w_lo", c_p * 4), ("ldw", i64 * 4),
                ("wt_hi", c_p * 4), ("wt_lo", c_p * 4), ("ldwt", i64 * 4)]


# name -> (restype, argtypes); mirrors include/egovlp_hip.h one to one (tests/test_abi.py checks it)
PROTOTYPES = {
    "egv_gemm_nt": (i32, [C.POINTER(GemmDesc), c_p]),
    "egv_split_f32": (i32, [c_p, i64, i32, i32, c_p, c_p, i64, c_p, c_p, i64, c_p, c_p]),
    "egv_transpose_planes": (i32, [c_p, c_p, i64, i32, i32, c_p, c_p, i64, c_p, c_p]),
    "egv_layernorm_fwd": (i32, [c_p, c_p, i64, c_p, c_p, f32, i32, i32, c_p, c_p, c_p, c_p, i64, c_p, c_p, c_p]),
    "egv_layernorm_bwd_parts": (i32, [i32]),
    "egv_layernorm_bwd": (i32, [c_p, c_p, c_p, i64, c_p, i64, c_p, c_p, c_p, i32, i32, c_p, c_p, c_p, i64, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_patch_gather": (i32, [c_p, i32, i32, i32, i32, i32, c_p, c_p, i64, c_p]),
    "egv_patch_gather_u8": (i32, [c_p, i32, i32, i32, i32, i32, c_p, c_p, c_p, c_p, i64, c_p]),
    "egv_patch_gather_u8_aug": (i32, [c_p, i32, i32, i32, i32, i32, i32, i32, c_p, c_p, c_p, c_p, c_p, i64, c_p]),
    "egv_assemble_tokens": (i32, [c_p, c_p, c_p, c_p, i32, i32, i32, i32, c_p, c_p]),
    "egv_assemble_tokens_bwd": (i32, [c_p, i32, i32, i32, i32, i32, c_p, c_p, c_p, c_p, c_p]),
    "egv_divided_attn_fwd": (i32, [c_p, c_p, i32, i32, i32, i32, i32, i32, c_p, c_p, c_p, c_p, c_p]),
    "egv_divided_attn_fwd_work_floats": (i64, [i32, i32, i32, i32, i32]),
    "egv_divided_attn_bwd": (i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, c_p, c_p, c_p, c_p]),
    "egv_divided_attn_bwd_work_floats": (i64, [i32, i32, i32, i32]),
    "egv_embed_fwd": (i32, [c_p, c_p, c_p, i32, i32, i32, c_p, c_p]),
    "egv_embed_bwd": (i32, [c_p, c_p, i32, i32, i32, i64, c_p, c_p, c_p]),
    "egv_text_attn_fwd": (i32, [c_p, c_p, c_p, i64, c_p, i32, i32, i32, i32, f32, u64, c_p, c_p, c_p, c_p, c_p]),
    "egv_text_attn_bwd": (i32, [c_p, c_p, c_p, i64, c_p, c_p, c_p, i32, i32, i32, i32, f32, u64, c_p, c_p, c_p, c_p, i64, c_p, c_p]),
    "egv_zero": (i32, [c_p, i64, c_p]),
    "egv_dropout": (i32, [c_p, c_p, c_p, i64, f32, u64, c_p, c_p]),
    "egv_egonce_fwd_bwd": (i32, [c_p, c_p, c_p, c_p, i32, i32, i32, i32, f32, f32, i32, i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_egonce_work_floats": (i64, [i32, i32]),
    "egv_sim_matrix_fwd": (i32, [c_p, c_p, i32, i32, i32, f32, c_p, c_p, c_p, c_p, c_p]),
    "egv_sim_matrix_bwd": (i32, [c_p, c_p, c_p, c_p, i32, i32, i32, f32, c_p, c_p, c_p]),
    "egv_egonce_from_sim": (i32, [c_p, c_p, c_p, i32, f32, i32, i32, c_p, c_p, c_p, c_p]),
    "egv_maxmargin_fwd_bwd": (i32, [c_p, c_p, i32, f32, i32, c_p, c_p, c_p]),
    "egv_dual_softmax": (i32, [c_p, i32, i32, f32, c_p, c_p, c_p]),
    "egv_cross_entropy_fwd_bwd": (i32, [c_p, i64, c_p, i32, i32, i64, c_p, c_p, i64, c_p]),
    "egv_adamw_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, f32, f32, f32, f32, f32, i32, i32, f32, c_p, c_p]),
    "egv_grad_pack_bf16": (i32, [i32, c_p, c_p, c_p, c_p, f32, c_p]),
    "egv_grad_unpack_bf16": (i32, [i32, c_p, c_p, c_p, c_p, c_p]),
    "egv_slice_sum_bf16": (i32, [c_p, i32, i64, c_p, c_p]),
    "egv_relu_split": (i32, [c_p, i64, i32, i32, c_p, c_p, i64, c_p]),
    "egv_version": (i32, []),
    "egv_abi_check": (i32, [i32, i64, i64, i64, i64]),
    "egv_split_f32_multi_t16": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_layernorm_bwd_fmt": (i32, [c_p, c_p, c_p, i64, c_p, i64, c_p, c_p, c_p, i32, i32, c_p, c_p, c_p, i64, c_p, c_p, i32, c_p, c_p, c_p, c_p]),
    "egv_layernorm_bwd_partial": (i32, [c_p, c_p, c_p, i64, c_p, i64, c_p, c_p, c_p, i32, i32, c_p, c_p, c_p, i64, c_p, c_p, i32, c_p, c_p, c_p, c_p]),
    "egv_layernorm_bwd_reduce": (i32, [i32, c_p, i32, i32, c_p, c_p, c_p]),
    "egv_splitk_reduce_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_grad_nonfinite_multi": (i32, [i32, c_p, c_p, c_p, c_p]),
    "egv_loss_scale_update": (i32, [c_p, c_p, f32, f32, f32, i32, i32, f32, f32, i32, f32, i32, c_p]),
    "egv_split_f32_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_f16x2_encode": (i32, [c_p, i64, i32, i32, c_p, c_p, c_p, i64, i32, c_p]),
    "egv_f16x2_encode_multi": (i32, [i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, c_p]),
    "egv_layernorm_fwd_f16x2": (i32, [c_p, i64, c_p, c_p, f32, i32, i32, c_p, c_p, c_p, i64, c_p, c_p, c_p]),
    "egv_block_fwd_arena_bytes": (i64, [C.POINTER(BlockGeom)]),
    "egv_block_fwd_offsets": (i32, [C.POINTER(BlockGeom), c_p]),
    "egv_block_fwd": (i32, [C.POINTER(BlockGeom), C.POINTER(BlockParams), c_p, c_p, c_p, c_p]),
    "egv_block_bwd_arena_bytes": (i64, [C.POINTER(BlockGeom), c_p]),
    "egv_block_grad_layout": (i32, [C.POINTER(BlockGeom), c_p, c_p]),
    "egv_block_bwd": (i32, [C.POINTER(BlockGeom), C.POINTER(BlockParams), C.POINTER(BlockBwdIO), c_p]),
    "egv_text_layer_fwd_arena_bytes": (i64, [C.POINTER(TextGeom)]),
    "egv_text_layer_fwd": (i32, [C.POINTER(TextGeom), C.POINTER(TextParams), c_p, c_p, c_p, c_p, c_p]),
    "egv_text_layer_bwd_arena_bytes": (i64, [C.POINTER(TextGeom)]),
    "egv_text_layer_grad_layout": (i32, [C.POINTER(TextGeom), c_p, c_p]),
    "egv_text_layer_bwd": (i32, [C.POINTER(TextGeom), C.POINTER(TextParams), c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "egv_diag_mfma_peak": (i32, [i32, i32, c_p, c_p]),
    "egv_diag_traffic_calib": (i32, [i32, c_p, c_p, i64, c_p]),
}

ABI_VERSION = 6      # EGV_ABI_VERSION of the header this binding was written against (include/egovlp_hip.h)
_lib = None


class EgovlpHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises EgovlpHipError if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EgovlpHipError(
                f"{LIB_PATH} not found: build it with `make -C egovlp_amd/csrc` (or __graft_entry__.build()). "
                "egovlp_amd has no CPU / eager fallback by design.")
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise EgovlpHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        # a library built from another header (struct layouts, mode / format codes) must fail here, not pass garbage in trailing fields
        if h.egv_abi_check(ABI_VERSION, C.sizeof(GemmDesc), C.sizeof(BlockGeom), C.sizeof(BlockParams), C.sizeof(BlockBwdIO)) != 0:
            raise EgovlpHipError(f"{LIB_PATH}: ABI mismatch (library version {h.egv_version()}, binding {ABI_VERSION}, or struct sizes "
                                 "differ): rebuild with `make -C egovlp_amd/csrc`")
        _lib = h
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        if rc == 1:
            raise EgovlpHipError(f"{what}: invalid argument (nothing was launched)")
        raise EgovlpHipError(f"{what}: HIP launch failed, hipError_t={rc - 2}")
