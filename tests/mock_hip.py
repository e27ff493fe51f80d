"""TEST INFRASTRUCTURE: a do-nothing stand-in for libegovlp_hip.so, so that the HOST side of the package (autograd
functions, execution contexts, stream / bucket bookkeeping, the training step, the trainer) can be driven end to end on CPU
tensors in this GPU-less container.  Every entry point of include/egovlp_hip.h is replaced by a ctypes callback with the
REAL prototype (egovlp_amd/_lib.PROTOTYPES), so argument counts and ctypes conversions are checked exactly as a real call
would check them; nothing is computed -- outputs are whatever torch.empty left there.  What such a dry run pins: arity and
types of every C-ABI call the step makes, shapes of every gradient the autograd functions return (autograd validates them
against the parameters), the order in which gradients become final (bucket launch order of the data-parallel exchange), and
the launch census of a step.  It says nothing about numerics: those are the -m gpu tests.

Never imported by the product (`egovlp_amd/`), by bench.py or by __graft_entry__.py.
"""
import contextlib
import ctypes as C


def _make_mock(calls):
    from egovlp_amd import _lib

    class Mock:
        pass

    m = Mock()
    m._keep = []
    special = {"egv_version": 3, "egv_layernorm_bwd_parts": 8, "egv_divided_attn_fwd_work_floats": 4096,
               "egv_divided_attn_bwd_work_floats": 4096, "egv_egonce_work_floats": 1 << 16}
    special.update({"egv_block_fwd_arena_bytes": 1 << 20, "egv_block_bwd_arena_bytes": 1 << 20,
                    "egv_text_layer_fwd_arena_bytes": 1 << 20, "egv_text_layer_bwd_arena_bytes": 1 << 20})

    def text_grad_layout(geom_p, off_p, tot_p):
        """egv_text_layer_grad_layout restated (csrc/text_layer.hip grad_layout)."""
        g = C.cast(geom_p, C.POINTER(_lib.TextGeom)).contents
        D, Hd = g.D, g.Hd
        shapes = [(3 * D, D), (D, D), (Hd, D), (D, Hd)]
        off = C.cast(off_p, C.POINTER(C.c_int64))
        p = k = 0
        for sizes in ([n * kk for n, kk in shapes], [n for n, _ in shapes], [D] * 4):
            for nel in sizes:
                off[k] = p
                p += nel
                k += 1
        C.cast(tot_p, C.POINTER(C.c_int64))[0] = p
        return 0

    def grad_layout(geom_p, off_p, tot_p):
        """egv_block_grad_layout restated (csrc/block.hip grad_layout): the host code slices its gradient views by these offsets."""
        g = C.cast(geom_p, C.POINTER(_lib.BlockGeom)).contents
        D, Hd = g.D, g.Hd
        shapes = [(3 * D, D), (D, D), (3 * D, D), (D, D), (Hd, D), (D, Hd)]
        off = C.cast(off_p, C.POINTER(C.c_int64))
        p = 0
        k = 0
        for sizes in ([n * kk for n, kk in shapes], [n for n, _ in shapes], [D] * 6):
            for nel in sizes:
                off[k] = p
                p += nel
                k += 1
        C.cast(tot_p, C.POINTER(C.c_int64))[0] = p
        return 0

    for name, (res, args) in _lib.PROTOTYPES.items():
        ret = special.get(name, 0)

        def cb(*a, _name=name, _ret=ret):
            calls.append(_name)
            if _name == "egv_block_fwd" and hasattr(calls, "block_single"):     # the per-block precision policy the host passed down
                calls.block_single.append(int(C.cast(a[0], C.POINTER(_lib.BlockGeom)).contents.f16_single))
            if _name == "egv_block_grad_layout":
                return grad_layout(*a)
            if _name == "egv_text_layer_grad_layout":
                return text_grad_layout(*a)
            return _ret

        proto = C.CFUNCTYPE(res, *args)
        fn = proto(cb)
        m._keep.append(fn)
        setattr(m, name, fn)
    return m


@contextlib.contextmanager
def mock_hip():
    """with mock_hip() as calls: ...  -- `calls` is the list of C-ABI entry points invoked, in order."""
    from egovlp_amd import _lib, ops

    class _Calls(list):
        """the call log; `.block_single`: egv_block_geom.f16_single of every egv_block_fwd call, in order"""

    calls = _Calls()
    calls.block_single = []
    saved = (_lib._lib, ops._stream, ops._need_cuda)
    _lib._lib = _make_mock(calls)
    ops._stream = lambda t=None: None
    ops._need_cuda = lambda *ts: None
    ops._SIZE_CACHE.clear()                  # workspace sizes answered by the mock must not outlive it
    try:
        yield calls
    finally:
        _lib._lib, ops._stream, ops._need_cuda = saved
        ops._SIZE_CACHE.clear()
