"""oracle/bst_oracle.py pinned against fixtures generated from the reference (tests/golden/make_golden_bst.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_bst as G                                   # case definitions + input generator (no reference import)
from oracle import bst_oracle as O


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "bst.npz"))


def _luts(lay, heads):
    return O.build_luts(lay)


@pytest.mark.parametrize("name", sorted(G.layouts().keys()))
def test_lookup_tables_and_masks_bit_exact(gold, name):
    lay = G.layouts()[name]
    L = O.build_luts(lay)
    for bsize in (8, 16, 32, 64):
        for cbn in ("causal", "checker"):
            key = "lut/%s/bs%d/%s/" % (name, bsize, cbn)
            if key + "nt_lut" not in gold.files:
                continue
            for t in ("nt_lut", "nn_lut", "tn_lut"):
                assert np.array_equal(L[t], gold[key + t]), (key, t)
            meta = gold[key + "meta"]
            assert [L["blocks"], L["nn_max"], L["tn_max"], L["lut_heads"], L["ctx_blks_q"], L["ctx_blks_k"]] == list(meta)
            m_np, m_k = O.build_masks(L, bsize, G.CALLBACKS[cbn])
            assert m_np.dtype == gold[key + "mask_np"].dtype
            assert np.array_equal(m_np, gold[key + "mask_np"]) and np.array_equal(m_k, gold[key + "mask"])


def _close(a, ref, tol, what):
    a, ref = G.sub(np.asarray(a, dtype=np.float64)), np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, what
    err = np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30)
    assert err < tol, (what, err)


@pytest.mark.parametrize("case", G.MATH_CASES, ids=[c[0] for c in G.MATH_CASES])
def test_math_matches_reference(gold, case):
    name, lkey, heads, bsize, hs, batch, cbn, seed = case
    lay = G.layouts()[lkey]
    L = O.build_luts(lay)
    inp = G.gen_inputs(lay, heads, bsize, hs, batch, L["blocks"], seed)
    key = "math/%s/" % name
    scale = float(gold[key + "scale"])
    _close(O.nt(L, inp["Q"], inp["K"], bsize, heads), gold[key + "NT"], 1e-6, "NT")
    _close(O.nn(L, inp["W"], inp["V"], bsize, heads), gold[key + "NN"], 1e-6, "NN")
    _close(O.tn(L, inp["W"], inp["E"], bsize, heads), gold[key + "TN"], 1e-6, "TN")
    mask_np = O.build_masks(L, bsize, G.CALLBACKS[cbn])[0] if cbn else None
    Y = O.masked_softmax(L, inp["X"], bsize, scale, mask_np)
    _close(Y, gold[key + "SM"], 1e-6, "SM")
    _close(O.masked_softmax_grad(L, inp["DY"], Y, scale), gold[key + "SMG"], 1e-6, "SMG")
    if key + "SM_AR" in gold.files:
        akey = int(gold[key + "akey"])
        m_k = O.build_masks(L, bsize, G.CALLBACKS[cbn])[1]
        m_ar = O.partial_autoregressive_mask(m_k, L["nt_lut"], bsize, akey)
        Yar = O.masked_softmax(L, inp["X"], bsize, scale, np.ascontiguousarray(m_ar.transpose(0, 2, 1)))
        _close(Yar, gold[key + "SM_AR"], 1e-6, "SM_AR")


def test_baseline_cfg5_layout_counts():
    lay = O.local_strided_layout(128)
    L = O.build_luts(lay)
    assert L["blocks"] == 1466 and L["nn_max"] == 19          # SURVEY.md section 8(d), cfg 5
