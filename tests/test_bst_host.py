"""CPU tier for the block-sparse attention path: product table builder and NumPy test methods against the fixtures
generated from the reference, and the C ABI of include/bst.h (symbols, struct layout, argument checks; no compute)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_bst as G


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "bst.npz"))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import _lib
    return _lib


@pytest.mark.parametrize("name", sorted(G.layouts().keys()))
def test_product_tables_and_masks_bit_exact(gold, name):
    from blocksparse_amd import BlocksparseTransformer
    lay = G.layouts()[name]
    heads = None if lay.ndim == 3 else 2
    for bsize in (8, 16, 32, 64):
        for cbn in ("causal", "checker"):
            key = "lut/%s/bs%d/%s/" % (name, bsize, cbn)
            if key + "nt_lut" not in gold.files:
                continue
            b = BlocksparseTransformer(lay, block_size=bsize, heads=heads, mask_callback=G.CALLBACKS[cbn])
            for t in ("nt_lut", "nn_lut", "tn_lut"):
                assert getattr(b, t).dtype == np.int32 and np.array_equal(getattr(b, t), gold[key + t]), (key, t)
            assert [b.blocks, b.nn_max, b.tn_max, b.lut_heads, b.ctx_blks_q, b.ctx_blks_k] == list(gold[key + "meta"])
            assert b.softmax_mask_np.dtype == gold[key + "mask_np"].dtype
            assert np.array_equal(b.softmax_mask_np, gold[key + "mask_np"]) and np.array_equal(b.softmax_mask, gold[key + "mask"])
            assert b.block_coord(0) == tuple(gold[key + "nt_lut"][0, 0])


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c[0] for c in G.MATH_CASES])
def test_numpy_test_methods_match_reference(gold, case):
    from blocksparse_amd import BlocksparseTransformer
    name, lkey, heads, bsize, hs, batch, cbn, seed = case
    lay = G.layouts()[lkey]
    b = BlocksparseTransformer(lay, block_size=bsize, heads=heads, mask_callback=G.CALLBACKS[cbn])
    inp = G.gen_inputs(lay, heads, bsize, hs, batch, b.blocks, seed)
    key = "math/%s/" % name
    scale = float(gold[key + "scale"])

    def close(a, ref, what):
        a = G.sub(a).astype(np.float64)
        err = np.linalg.norm(a - ref) / np.linalg.norm(ref)
        assert err < 2e-6, (what, err)

    close(b.nt_test(inp["Q"], inp["K"]), gold[key + "NT"], "NT")
    close(b.nn_test(inp["W"], inp["V"]), gold[key + "NN"], "NN")
    close(b.tn_test(inp["W"], inp["E"]), gold[key + "TN"], "TN")
    Y = b.masked_softmax_test(inp["X"], scale=scale)
    close(Y, gold[key + "SM"], "SM")
    close(b.masked_softmax_grad_test(inp["DY"], Y, scale=scale), gold[key + "SMG"], "SMG")
    if key + "SM_AR" in gold.files:
        close(b.masked_softmax_test(inp["X"], scale=scale, autoregress_at_key=int(gold[key + "akey"])), gold[key + "SM_AR"], "SM_AR")


def test_constructor_rejections():
    from blocksparse_amd import BlocksparseTransformer
    with pytest.raises(AssertionError):
        BlocksparseTransformer(np.ones((4, 4)), block_size=32)                       # shared layout needs heads
    with pytest.raises(AssertionError):
        BlocksparseTransformer(np.ones((2, 4, 4)), block_size=12)
    lay = np.ones((2, 4, 4), dtype=np.int32)
    lay[1, 0, 0] = 0
    with pytest.raises(AssertionError):
        BlocksparseTransformer(lay, block_size=32)                                   # unequal block counts across heads


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bst.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                     # the header comment cites reference launchers
    declared = set(re.findall(r"\b(bst_[a-z_]+)\s*\(", code)) - {"bst_args"}
    assert declared == set(lib.BST_SYMBOLS), declared ^ set(lib.BST_SYMBOLS)
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in declared:
        getattr(raw, s)


def test_struct_layout_matches_header(lib):
    hdr = open(os.path.join(ROOT, "include", "bst.h")).read()
    body = re.search(r"typedef struct bst_args \{(.*?)\} bst_args;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.search(r"(\w+)\s*;", ln).group(1) for ln in body.splitlines() if ";" in ln]
    assert names == [f[0] for f in lib.BstArgs._fields_]
    assert ctypes.sizeof(lib.BstArgs) == 8 + 11 * 4 + 4 + 8


def test_argument_validation_without_gpu(lib):
    L = lib.load()
    one = ctypes.c_void_p(256)
    assert L.bst_nt(one, one, one, None) == -1
    a = lib.BstArgs()
    a.lut = 256
    a.lut_heads, a.lut_dim, a.blocks, a.bsize, a.batch, a.heads, a.head_state = 1, 10, 10, 32, 2, 4, 64
    a.ctx_blks_q, a.ctx_blks_k, a.dtype, a.score_dtype = 4, 4, lib.F32, lib.BF16
    assert L.bst_nt(None, one, one, ctypes.byref(a)) == -1
    a.bsize = 12
    assert L.bst_nt(one, one, one, ctypes.byref(a)) == -2
    a.bsize, a.head_state = 32, 60                                 # not a multiple of 8 (src/bst_op.cc:208)
    assert L.bst_nt(one, one, one, ctypes.byref(a)) == -1
    a.head_state, a.lut_heads = 64, 3                              # neither 1 nor heads (src/bst_op.cc:209)
    assert L.bst_nn(one, one, one, ctypes.byref(a)) == -1
    a.lut_heads, a.score_dtype = 1, lib.F32
    assert L.bst_nn(one, one, one, ctypes.byref(a)) == -2
    a.score_dtype = lib.BF16                                       # nn expects lut_dim == ctx_blks_q + blocks
    assert L.bst_nn(one, one, one, ctypes.byref(a)) == -1
    a.lut_dim = 14
    assert L.bst_nt(one, one, one, ctypes.byref(a)) == -1          # nt expects lut_dim == blocks
    assert L.bst_masked_softmax(one, one, None, 1, 1.0, lib.F32, lib.BF16, ctypes.byref(a)) == -2
    assert L.bst_masked_softmax(one, one, one, 3, 1.0, lib.BF16, lib.BF16, ctypes.byref(a)) == -1
    assert L.bst_softmax_grad(one, one, one, 1.0, lib.F32, ctypes.byref(a)) == -2
    assert L.bst_partial_autoregressive_mask(one, one, one, 32, 10, 1, -1, None) == -1
    assert L.bst_partial_autoregressive_mask(one, one, one, 24, 10, 1, 5, None) == -2
