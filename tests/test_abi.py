"""CPU tier: the C-ABI library loads, exports every symbol include/bsmm.h declares, and rejects bad
arguments before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    declared = set(re.findall(r"\b(bsmm_[a-z_]+)\s*\(", hdr))
    declared -= {"bsmm_args", "bsmm_params"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    L = lib.load()
    for s in declared:
        assert hasattr(L, s), s
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in declared:
        getattr(raw, s)


def test_version_and_error_strings(lib):
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    assert L.bsmm_version() == int(re.search(r"#define BSMM_VERSION (\d+)", hdr).group(1))
    assert lib.error_string(0) == "ok"
    for code in (-1, -2, -3):
        assert "bsmm" in lib.error_string(code)


def test_struct_layout_matches_header(lib):
    # field order in the ctypes mirror == field order in the C struct
    hdr = open(os.path.join(ROOT, "include", "bsmm.h")).read()
    body = re.search(r"typedef struct bsmm_args \{(.*?)\} bsmm_args;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.search(r"(\w+)\s*;", ln).group(1) for ln in body.splitlines() if ";" in ln]
    assert names == [f[0] for f in lib.BsmmArgs._fields_]
    assert ctypes.sizeof(lib.BsmmArgs) == 4 * 8 + 11 * 4 + 2 * 4 + 4 + 8   # 3 ptr + size_t, 11 int32, 2 float, pad, ptr


def test_argument_validation_without_gpu(lib):
    L = lib.load()
    a = lib.BsmmArgs()
    one = ctypes.c_void_p(256)          # a non-null, 16-byte aligned dummy address: never dereferenced
    assert L.bsmm_fprop(one, one, one, None) == -1
    a.lut = 256
    a.blocks, a.N, a.C, a.K, a.segments = 4, 8, 64, 64, 2
    a.bsize = 7
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -2
    a.bsize, a.axis = 32, 3
    assert L.bsmm_bprop(one, one, one, ctypes.byref(a)) == -2
    a.axis, a.dtype = 0, 9
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -2
    a.dtype = lib.BF16
    a.gate = 256
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -2
    a.gate = None
    a.C = 65                                     # not a multiple of bsize
    assert L.bsmm_fprop(one, one, one, ctypes.byref(a)) == -1
    a.C = 64
    a.pcount = 9
    arr = (ctypes.c_void_p * 1)(256)
    assert L.bsmm_updat(arr, arr, one, ctypes.byref(a)) == -1
    # workspace query is pure host arithmetic
    a.blocks, a.bsize, a.dtype = 10, 32, lib.BF16
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 10 * 32 * 32 * 2
    assert L.bsmm_workspace_bytes(lib.OP_BPROP, ctypes.byref(a)) == 0
    a.bsize = 8
    assert L.bsmm_workspace_bytes(lib.OP_FPROP, ctypes.byref(a)) == 0


def test_host_class_surface():
    import numpy as np
    from blocksparse_amd import BlocksparseMatMul
    lay = np.array([[1, 0, 1], [1, 1, 0]])
    b = BlocksparseMatMul(lay, block_size=16, feature_axis=1)
    assert b.w_shape == (4, 16, 16) and b.C == 32 and b.K == 48 and b.blocks == 4
    assert b.i_shape(5) == (5, 32) and b.o_shape(5) == (5, 48) and b.flops == 4 * 16 * 16 * 2
    assert b.block_coord(0) == tuple(b.updat_lut[0])
    for bad in ((64, 0), (32, 2), (4, 0)):
        with pytest.raises(ValueError):
            BlocksparseMatMul(lay, block_size=bad[0], feature_axis=bad[1])
    import pickle
    b2 = pickle.loads(pickle.dumps(b))
    np.testing.assert_array_equal(b2.fprop_lut, b.fprop_lut)
    # host NumPy reference methods agree with the oracle
    from oracle import bsmm_oracle as orc
    t = orc.build_layout_luts(lay, 16)
    rng = np.random.default_rng(0)
    W = rng.normal(size=b.w_shape); X = rng.normal(size=b.i_shape(6)); E = rng.normal(size=b.o_shape(6))
    np.testing.assert_allclose(b.fprop_test(X, W), orc.fprop(t, X, W, 1), atol=1e-12)
    np.testing.assert_allclose(b.bprop_test(E, W), orc.bprop(t, E, W, 1), atol=1e-12)
    np.testing.assert_allclose(b.updat_test(X, E), orc.updat(t, X, E, 1), atol=1e-12)
    b0 = BlocksparseMatMul(lay, block_size=16, feature_axis=0)
    np.testing.assert_allclose(b0.fprop_test(X.T.copy(), W), orc.fprop(t, X.T.copy(), W, 0), atol=1e-12)
    np.testing.assert_allclose(b0.bprop_test(E.T.copy(), W), orc.bprop(t, E.T.copy(), W, 0), atol=1e-12)
    np.testing.assert_allclose(b0.updat_test(X.T.copy(), E.T.copy()), orc.updat(t, X.T.copy(), E.T.copy(), 0), atol=1e-12)


def test_cpu_tensors_are_rejected_loudly():
    import numpy as np
    import torch
    from blocksparse_amd import BlocksparseMatMul
    b = BlocksparseMatMul(np.ones((2, 2)), block_size=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        b.fprop(torch.zeros(b.i_shape(4)), torch.zeros(b.w_shape))
