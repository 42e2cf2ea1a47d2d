"""ctypes binding of libbsmm_hip.so (the C ABI declared in include/bsmm.h).

There is NO CPU fallback: if the shared library is missing this module raises, loudly, at first use.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``blocksparse_amd.build.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSMM_LIB: load another build of the same library (kernel A/B experiments: scripts/build_variants.py); product = the default
LIB_PATH = os.environ.get("BSMM_LIB") or os.path.join(_HERE, "libbsmm_hip.so")

ABI_VERSION = 128        # include/bsmm.h BSMM_VERSION this binding was written against (struct layout, plan formats, option bits)
F32, F16, BF16 = 0, 1, 2
OP_FPROP, OP_BPROP, OP_UPDAT = 0, 1, 2
FLAG_GATED_DW, FLAG_FORCE_VALU, FLAG_NO_PLAN, FLAG_FORCE_PLAN, FLAG_DW_SUMS, FLAG_FORCE_MID = 1, 2, 4, 8, 16, 32
# bsmm_args.trace codes (include/bsmm.h BSMM_K_*)
K_XPROP_VALU, K_XPROP_SEGMENT, K_XCOL32, K_XCOL16, K_XCOL32_F32SPLIT, K_XCOL32_F32MFMA, K_XPROP_SUPER8 = 1, 2, 3, 4, 5, 6, 7
K_XCOL32_STAGED = 8
K_XCOL16_STAGED = 9
K_XCOL32_FLOW = 10
K_XPROP_SMALL = 11
K_XPROP_MID = 12
K_XCOL32_ROWS = 13
KV_ONE_WAVE = 1          # trace variant of K_UPDAT_BLOCK_TR: the small-minibatch form, one wave per block
KV_FLOW_HALF_UNITS = 2   # trace variant of K_XCOL32_FLOW: units of 64 rows (0 = the 128-row units the bench times)
K_UPDAT_VALU, K_UPDAT_BLOCK, K_UPDAT_BLOCK_TR, K_UPDAT_WIN, K_UPDAT16_WIN, K_UPDAT_SUPER8, K_UPDAT_STREAM, K_UPDAT16_ROWS = 16, 17, 18, 19, 20, 21, 22, 23
# plan-builder options (BSMM_PLAN_*)
PLAN_XCOL_UNSTAGED = 4
PLAN_XCOL_FLOW = 8
PLAN_XPROP_PH_SHIFT, PLAN_UPDAT_SETS_SHIFT = 8, 12
PLAN_UPDAT16_WINDOWED = 0x40000  # bsize 16, feature axis 0: no 'BSU6' section (always the windowed weight-gradient kernel)
PLAN_UPDAT_NO_DIRECT = 0x80000   # 'BSU2' plans without direct blocks (round 6): overflow items in a sliced last round instead
PLAN_XCOL_ROWS = 0x20000       # retired in round 6 (the row-split kernel of round 5): ignored by the builders
PLAN_FLOW_CONSECUTIVE = 0x100000  # BSX4 plans: never regroup the output blocks of an unbalanced layout (round 6)
PLAN_FLOW_SCHEDULED = 0x10000  # BSX4 plans, experiment: list-scheduled step order instead of ascending input blocks
PLAN_WINDOW_MASK = 0xf0         # the window / kernel-family field of the updat plan options
PLAN_XCOL_NARROW, PLAN_F32_MFMA, PLAN_WINDOW_8, PLAN_WINDOW_16, PLAN_WINDOW_16W, PLAN_STREAM_16, PLAN_STREAM_8, PLAN_STREAM_32 = 1, 2, 0x10, 0x20, 0x30, 0x40, 0x50, 0x60

SYMBOLS = ("bsmm_fprop", "bsmm_bprop", "bsmm_updat", "bsmm_updat_finalize", "bsmm_identity_init", "bsmm_gate_grad", "bsmm_gate_weights", "bsmm_l2_normalize", "bsmm_l2_normalize_grad", "bsmm_sparse_op", "bsmm_sparse_mul_grad", "bsmm_workspace_bytes",
           "bsmm_xprop_plan_words", "bsmm_xprop_plan_build", "bsmm_updat_plan_words", "bsmm_updat_plan_build",
           "bsmm_plan_attach", "bsmm_error_string", "bsmm_version", "bsmm_prepared_bytes", "bsmm_prepare_weights")
DIST_SYMBOLS = ("bsmm_dist_unique_id", "bsmm_dist_create", "bsmm_dist_allreduce_begin", "bsmm_dist_allreduce_end", "bsmm_dist_stream",
                "bsmm_dist_world", "bsmm_dist_destroy", "bsmm_dist_dw_shard_elems", "bsmm_dist_dw_layout", "bsmm_dist_dw_begin", "bsmm_dist_dw_emulate",
                "bsmm_dist_dw_end")
BST_SYMBOLS = ("bst_nt", "bst_nn", "bst_tn", "bst_masked_softmax", "bst_softmax_grad", "bst_partial_autoregressive_mask", "bst_nt_softmax", "bst_nt_softmax_grad")


class BsmmArgs(ctypes.Structure):
    """Mirror of ``struct bsmm_args`` (include/bsmm.h)."""
    _fields_ = [
        ("lut", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_size_t), ("plan", ctypes.c_void_p),
        ("plan_magic", ctypes.c_int32), ("plan_width", ctypes.c_int32), ("plan_waves", ctypes.c_int32),
        ("plan_items", ctypes.c_int32), ("plan_inner", ctypes.c_int32), ("flags", ctypes.c_int32), ("split", ctypes.c_int32),
        ("blocks", ctypes.c_int32), ("bsize", ctypes.c_int32), ("segments", ctypes.c_int32),
        ("locks", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
        ("shared", ctypes.c_int32), ("pcount", ctypes.c_int32), ("axis", ctypes.c_int32),
        ("dtype", ctypes.c_int32), ("alpha", ctypes.c_float), ("beta", ctypes.c_float),
        ("stream", ctypes.c_void_p), ("trace", ctypes.POINTER(ctypes.c_int32)), ("prepared_w", ctypes.c_void_p),
    ]


class BstArgs(ctypes.Structure):
    """Mirror of ``struct bst_args`` (include/bst.h)."""
    _fields_ = [
        ("lut", ctypes.c_void_p), ("lut_heads", ctypes.c_int32), ("lut_dim", ctypes.c_int32),
        ("blocks", ctypes.c_int32), ("bsize", ctypes.c_int32), ("batch", ctypes.c_int32), ("heads", ctypes.c_int32),
        ("head_state", ctypes.c_int32), ("ctx_blks_q", ctypes.c_int32), ("ctx_blks_k", ctypes.c_int32),
        ("dtype", ctypes.c_int32), ("score_dtype", ctypes.c_int32), ("flags", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


_lib = None

# Test hook (host side only -- the library itself keeps no switches): flags OR-ed into every call the host classes make.
#   0 production dispatch, 1 = FLAG_FORCE_VALU, 2 = FLAG_NO_PLAN, 3 = FLAG_FORCE_PLAN, 4 = FLAG_FORCE_MID
_VARIANT_FLAGS = {0: 0, 1: FLAG_FORCE_VALU, 2: FLAG_NO_PLAN, 3: FLAG_FORCE_PLAN, 4: FLAG_FORCE_MID}
_call_flags = 0
_last_kernel = ctypes.c_int32(0)


def set_kernel_variant(variant):
    global _call_flags
    _call_flags = _VARIANT_FLAGS.get(int(variant), 0)


def call_flags():
    return _call_flags


def last_kernel():
    """BSMM_K_* code of the kernel family the most recent fprop / bprop / updat call of the host classes dispatched to."""
    return int(_last_kernel.value) & 0xff


def last_kernel_variant():
    """BSMM_KV_* variant inside that family (bits 8..15 of the trace word; 0 = the plain one)."""
    return (int(_last_kernel.value) >> 8) & 0xff


class BsmmError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed with code %d: %s" % (where, code, error_string(code)))


def load():
    """Load (once) and return the ctypes handle; raise if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "blocksparse_amd: %s not found -- the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` in the repo root. "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    pargs = ctypes.POINTER(BsmmArgs)
    lib.bsmm_fprop.argtypes = [vp, vp, vp, pargs]
    lib.bsmm_fprop.restype = ctypes.c_int
    lib.bsmm_bprop.argtypes = [vp, vp, vp, pargs]
    lib.bsmm_bprop.restype = ctypes.c_int
    lib.bsmm_updat.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), vp, pargs]
    lib.bsmm_updat.restype = ctypes.c_int
    lib.bsmm_updat_finalize.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32, vp]
    lib.bsmm_updat_finalize.restype = ctypes.c_int
    lib.bsmm_dist_unique_id.argtypes = [vp]
    lib.bsmm_dist_unique_id.restype = ctypes.c_int
    lib.bsmm_dist_create.argtypes = [ctypes.POINTER(vp), vp, i32, i32, i32]
    lib.bsmm_dist_create.restype = ctypes.c_int
    lib.bsmm_dist_allreduce_begin.argtypes = [vp, vp, ctypes.c_size_t, i32, vp]
    lib.bsmm_dist_allreduce_begin.restype = ctypes.c_int
    lib.bsmm_dist_allreduce_end.argtypes = [vp, vp]
    lib.bsmm_dist_allreduce_end.restype = ctypes.c_int
    lib.bsmm_dist_stream.argtypes = [vp]
    lib.bsmm_dist_stream.restype = vp
    lib.bsmm_dist_world.argtypes = [vp]
    lib.bsmm_dist_world.restype = ctypes.c_int
    lib.bsmm_dist_destroy.argtypes = [vp]
    lib.bsmm_dist_destroy.restype = ctypes.c_int
    lib.bsmm_dist_dw_shard_elems.argtypes = [i32, i32, i32]
    lib.bsmm_dist_dw_shard_elems.restype = ctypes.c_size_t
    psz = ctypes.POINTER(ctypes.c_size_t)
    lib.bsmm_dist_dw_layout.argtypes = [i32, i32, i32, i32, psz, psz, psz, psz]
    lib.bsmm_dist_dw_layout.restype = ctypes.c_int
    lib.bsmm_dist_dw_begin.argtypes = [vp, vp, ctypes.c_size_t, vp, vp, vp, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp]
    lib.bsmm_dist_dw_begin.restype = ctypes.c_int
    lib.bsmm_dist_dw_emulate.argtypes = [i32, ctypes.POINTER(vp), ctypes.c_size_t, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, i32, i32, i32,
                                         ctypes.c_float, ctypes.c_float, vp]
    lib.bsmm_dist_dw_emulate.restype = ctypes.c_int
    lib.bsmm_dist_dw_end.argtypes = [vp, vp]
    lib.bsmm_dist_dw_end.restype = ctypes.c_int
    lib.bsmm_identity_init.argtypes = [vp, vp, i32, i32, i32, i32, f32, i32, vp]
    lib.bsmm_identity_init.restype = ctypes.c_int
    lib.bsmm_gate_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.bsmm_gate_grad.restype = ctypes.c_int
    lib.bsmm_gate_weights.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.bsmm_gate_weights.restype = ctypes.c_int
    lib.bsmm_l2_normalize.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.bsmm_l2_normalize.restype = ctypes.c_int
    lib.bsmm_l2_normalize_grad.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.bsmm_l2_normalize_grad.restype = ctypes.c_int
    lib.bsmm_sparse_op.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.bsmm_sparse_op.restype = ctypes.c_int
    lib.bsmm_sparse_mul_grad.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.bsmm_sparse_mul_grad.restype = ctypes.c_int
    lib.bsmm_workspace_bytes.argtypes = [ctypes.c_int, pargs]
    lib.bsmm_workspace_bytes.restype = ctypes.c_size_t
    lib.bsmm_prepared_bytes.argtypes = [ctypes.c_int, pargs]
    lib.bsmm_prepared_bytes.restype = ctypes.c_size_t
    lib.bsmm_prepare_weights.argtypes = [ctypes.c_int, vp, vp, pargs]
    lib.bsmm_prepare_weights.restype = ctypes.c_int
    ip = ctypes.POINTER(ctypes.c_int32)
    lib.bsmm_xprop_plan_words.argtypes = [ip, i32, i32, i32, i32, i32, i32, i32]
    lib.bsmm_xprop_plan_words.restype = ctypes.c_long
    lib.bsmm_xprop_plan_build.argtypes = [ip, i32, i32, i32, i32, i32, i32, i32, ip]
    lib.bsmm_xprop_plan_build.restype = ctypes.c_int
    lib.bsmm_updat_plan_words.argtypes = [ip, i32, i32, i32, i32, i32, i32, i32]
    lib.bsmm_updat_plan_words.restype = ctypes.c_long
    lib.bsmm_updat_plan_build.argtypes = [ip, i32, i32, i32, i32, i32, i32, i32, ip]
    lib.bsmm_updat_plan_build.restype = ctypes.c_int
    lib.bsmm_plan_attach.argtypes = [pargs, ip, ctypes.c_long, vp]
    lib.bsmm_plan_attach.restype = ctypes.c_int
    lib.bsmm_error_string.argtypes = [ctypes.c_int]
    lib.bsmm_error_string.restype = ctypes.c_char_p
    lib.bsmm_version.argtypes = []
    lib.bsmm_version.restype = ctypes.c_int
    pbst = ctypes.POINTER(BstArgs)
    for name in ("bst_nt", "bst_nn", "bst_tn"):
        getattr(lib, name).argtypes = [vp, vp, vp, pbst]
        getattr(lib, name).restype = ctypes.c_int
    lib.bst_masked_softmax.argtypes = [vp, vp, vp, i32, f32, i32, i32, pbst]
    lib.bst_masked_softmax.restype = ctypes.c_int
    lib.bst_nt_softmax.argtypes = [vp, vp, vp, vp, i32, f32, i32, pbst]
    lib.bst_nt_softmax.restype = ctypes.c_int
    lib.bst_nt_softmax_grad.argtypes = [vp, vp, vp, vp, f32, i32, pbst]
    lib.bst_nt_softmax_grad.restype = ctypes.c_int
    lib.bst_softmax_grad.argtypes = [vp, vp, vp, f32, i32, pbst]
    lib.bst_softmax_grad.restype = ctypes.c_int
    lib.bst_partial_autoregressive_mask.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.bst_partial_autoregressive_mask.restype = ctypes.c_int
    if lib.bsmm_version() != ABI_VERSION:
        raise RuntimeError("blocksparse_amd: %s reports ABI version %d, this binding expects %d -- rebuild the library "
                           "(python -c 'import __graft_entry__ as g; g.build()')" % (LIB_PATH, lib.bsmm_version(), ABI_VERSION))
    _lib = lib
    return lib


def dw_layout(world, rank, blocks, bsize):
    """(shard, lo, hi, capacity) of the fused dw reduction for one rank (bsmm_dist_dw_layout: host arithmetic, no device needed)"""
    sh, lo, hi, cap = (ctypes.c_size_t() for _ in range(4))
    check(load().bsmm_dist_dw_layout(world, rank, blocks, bsize, ctypes.byref(sh), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(cap)),
          "bsmm_dist_dw_layout")
    return int(sh.value), int(lo.value), int(hi.value), int(cap.value)


def error_string(code):
    return load().bsmm_error_string(int(code)).decode()


def check(code, where):
    if code != 0:
        raise BsmmError(code, where)


def raw_stream(device):
    """The current HIP stream of ``device`` as an integer handle.  ``torch.cuda.current_stream(dev).cuda_stream`` builds a Stream object on every
    call (4 us of the ~12 us an eager small-minibatch call spends on the host, scripts/gpu_host_overhead.py); the C binding returns the handle."""
    import torch
    idx = getattr(device, "index", None) if isinstance(device, torch.device) else torch.device(device).index     # (a device may arrive as "cuda" / "cuda:1")
    if idx is None:
        idx = torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:          # (an older PyTorch)
        return torch.cuda.current_stream(device).cuda_stream
