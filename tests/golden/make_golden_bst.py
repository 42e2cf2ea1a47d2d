"""Generate tests/golden/bst.npz FROM THE REFERENCE ITSELF (block-sparse attention path, BASELINE configs[4]).

Authoring container only (needs /root/reference):    python tests/golden/make_golden_bst.py

Imports /root/reference/blocksparse/transformer.py behind a fake ``tensorflow`` module (only the NumPy table builder
and the NumPy ``*_test`` oracles of BlocksparseTransformer run) and stores, for a few seeded cases, the lookup tables,
packed softmax masks and the outputs of nt_test / nn_test / tn_test / masked_softmax_test / masked_softmax_grad_test.
Inputs are not stored: tests regenerate them from the recorded RandomState seeds (``gen_inputs`` below is imported by
the tests).  The reference file is executed unmodified.
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference_transformer():
    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class Any:
        def __getattr__(self, k):
            return Any()

        def __call__(self, *a, **k):
            return Any()

    tf = mod("tensorflow")
    tf.__getattr__ = lambda k: Any()
    mod("tensorflow.python")
    mod("tensorflow.python.framework")
    mod("tensorflow.python.framework.ops", RegisterGradient=lambda name: (lambda fn: fn))
    pkg = mod("blocksparse")
    pkg.__path__ = [os.path.join(REF, "blocksparse")]
    mod("blocksparse.utils", _op_module=Any(), scalar_constant=lambda *a, **k: None)
    sys.path.insert(0, REF)
    if not hasattr(np, "bool"):
        np.bool = bool                       # the reference predates NumPy 1.24
    return importlib.import_module("blocksparse.transformer")


# ---- case definitions (shared with the tests) ------------------------------------------------------
def causal_cb(blk_shape, h, q, k, b):
    m = np.ones(blk_shape, dtype=bool)
    if q == k:
        m = np.tril(m)
    return m


def checker_cb(blk_shape, h, q, k, b):
    m = np.ones(blk_shape, dtype=bool)
    m[::2, 1::2] = False
    m[1::2, ::2] = False
    return m


def head_cb(blk_shape, h, q, k, b):          # differs per head and per block
    rs = np.random.RandomState(1000 * h + b)
    m = rs.rand(*blk_shape) < 0.7
    m[:, 0] = True                           # never a fully masked row
    return m


def layouts():
    out = {}
    tri = np.tril(np.ones((8, 8), dtype=np.int32))
    out["causal_shared"] = tri                                            # 2-D: shared by all heads
    out["causal_2heads"] = np.stack([tri, tri])
    q, k = np.indices((32, 32))
    out["local_strided_32"] = ((k <= q) & ((q - k < 4) | ((q - k) % 8 == 0))).astype(np.int32)
    rs = np.random.RandomState(7)
    per = []
    for h in range(3):                                                    # per-head layouts, equal block count, Qb != Kb
        m = np.zeros(6 * 10, dtype=np.int32)
        m[rs.permutation(60)[:23]] = 1
        per.append(m.reshape(6, 10))
    out["rect_3heads"] = np.stack(per)
    out["dense_4x4"] = np.ones((4, 4), dtype=np.int32)
    return out


MATH_CASES = [
    # name, layout key, heads, bsize, head_state, batch, mask callback name, seed
    ("causal32", "causal_2heads", 2, 32, 64, 2, "causal", 11),
    ("shared32", "causal_shared", 4, 32, 16, 1, "causal", 12),
    ("rect16", "rect_3heads", 3, 16, 8, 2, "head", 13),
    ("dense8", "dense_4x4", 2, 8, 8, 2, "checker", 14),
    ("rect64", "rect_3heads", 3, 64, 8, 1, None, 15),
]
CALLBACKS = {"causal": causal_cb, "checker": checker_cb, "head": head_cb, None: None}


def gen_inputs(layout, heads, bsize, head_state, batch, blocks, seed):
    """fp16-representable inputs as the reference test draws them (test/blocksparse_transformer_test.py:133-136)."""
    rs = np.random.RandomState(seed)
    lay = layout if layout.ndim == 3 else layout[None]
    Qb, Kb = lay.shape[1:]
    f = lambda *s: rs.uniform(-1.0, 1.0, s).astype(np.float16).astype(np.float32)
    return dict(Q=f(batch, Qb * bsize, heads * head_state), K=f(batch, Kb * bsize, heads * head_state),
                V=f(batch, Kb * bsize, heads * head_state), E=f(batch, Qb * bsize, heads * head_state),
                W=f(batch, heads, blocks, bsize, bsize),
                X=rs.normal(0.0, 1.0, (batch, heads, blocks, bsize, bsize)).astype(np.float16).astype(np.float32),
                DY=rs.normal(0.0, 1.0, (batch, heads, blocks, bsize, bsize)).astype(np.float16).astype(np.float32))


SUB_LIMIT = 30000


def sub(a):
    """The fixture keeps every ceil(size / SUB_LIMIT)-th element of a large array (flattened); tests apply the same."""
    a = np.asarray(a)
    if a.size <= SUB_LIMIT:
        return a
    return a.reshape(-1)[:: -(-a.size // SUB_LIMIT)].copy()


def main():
    tr = import_reference_transformer()
    out = {}
    lays = layouts()
    for name, lay in lays.items():
        heads = None if lay.ndim == 3 else 2
        for bsize in (8, 16, 32, 64):
            for cbn in ("causal", "checker"):
                if cbn == "checker" and bsize != 32:
                    continue
                b = tr.BlocksparseTransformer(lay, block_size=bsize, heads=heads, mask_callback=CALLBACKS[cbn])
                key = "lut/%s/bs%d/%s/" % (name, bsize, cbn)
                out[key + "nt_lut"], out[key + "nn_lut"], out[key + "tn_lut"] = b.nt_lut, b.nn_lut, b.tn_lut
                out[key + "meta"] = np.array([b.blocks, b.nn_max, b.tn_max, b.lut_heads, b.ctx_blks_q, b.ctx_blks_k], dtype=np.int64)
                out[key + "mask_np"], out[key + "mask"] = b.softmax_mask_np, b.softmax_mask
    for name, lkey, heads, bsize, hs, batch, cbn, seed in MATH_CASES:
        lay = lays[lkey]
        b = tr.BlocksparseTransformer(lay, block_size=bsize, heads=heads, mask_callback=CALLBACKS[cbn])
        inp = gen_inputs(lay, heads, bsize, hs, batch, b.blocks, seed)
        key = "math/%s/" % name
        scale = 1.0 / np.sqrt(hs)
        out[key + "scale"] = np.float64(scale)
        out[key + "NT"] = b.nt_test(inp["Q"], inp["K"])
        out[key + "NN"] = b.nn_test(inp["W"], inp["V"])
        out[key + "TN"] = b.tn_test(inp["W"], inp["E"])
        if cbn is not None:
            Y = b.masked_softmax_test(inp["X"], scale=scale)
            out[key + "SM"] = Y
            if cbn == "causal":
                akey = 3 * bsize + bsize // 2 - 1
                out[key + "akey"] = np.int64(akey)
                out[key + "SM_AR"] = b.masked_softmax_test(inp["X"], scale=scale, autoregress_at_key=akey)
        else:
            b.softmax_mask_np = None
            Y = b.masked_softmax_test(inp["X"], scale=scale)
            out[key + "SM"] = Y
        out[key + "SMG"] = b.masked_softmax_grad_test(inp["DY"], Y, scale=scale)
    # large outputs are stored as a strided subsample of the flattened array (see ``sub``): random data does not compress
    for k in list(out):
        if k.startswith("math/") and out[k].size > SUB_LIMIT:
            out[k] = sub(out[k])
    path = os.path.join(HERE, "bst.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
