"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports /root/reference/blocksparse/matmul.py behind a fake ``tensorflow`` module (the reference is a
TF-1.x op library; only its NumPy LUT builder and NumPy ``*_test`` oracles are executed here) and stores
  * the lookup tables the reference builds for several layouts, and
  * inputs + outputs of the reference's fprop_test / bprop_test / updat_test for small seeded cases
as compressed .npz files.  Tests then compare (a) oracle/bsmm_oracle.py, (b) the product LUT builder and
(c) the HIP kernels against these files; nothing at test time reads /root/reference.

One deliberate patch: ``scipy.sparse.find`` is wrapped to return entries column-major, which is what the
reference's builder assumes (blocksparse/matmul.py:113-117, "ks is in sorted order by default") and what
the SciPy of its era did; SciPy 1.15 returns row-major (SURVEY TRAP 3).  Nothing else is modified.
"""
import importlib
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference_matmul():
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
    tf.resource_loader = types.SimpleNamespace(get_data_files_path=lambda: "/nonexistent")
    tf.load_op_library = lambda path: Any()
    mod("tensorflow.python")
    mod("tensorflow.python.framework")
    mod("tensorflow.python.ops")
    mod("tensorflow.python.framework.ops", RegisterGradient=lambda name: (lambda fn: fn))
    mod("tensorflow.python.ops.init_ops", Initializer=object)
    pkg = mod("blocksparse")
    pkg.__path__ = [os.path.join(REF, "blocksparse")]
    mod("blocksparse.ewops")
    sys.path.insert(0, REF)
    mm = importlib.import_module("blocksparse.matmul")
    _find = sp.find

    def find_colmajor(A):
        r, c, v = _find(A)
        o = np.lexsort((r, c))
        return r[o], c[o], v[o]

    mm.sparse.find = find_colmajor
    return mm


def ba_layout(n, m, seed, dense_corner=True):
    import networkx
    g = networkx.generators.barabasi_albert_graph(n, m, seed=seed)
    lay = networkx.adjacency_matrix(g).toarray().astype(np.int32) + np.eye(n, dtype=np.int32)
    if dense_corner:
        lay[0:m, 0:m] = 1
    return (lay != 0).astype(np.int32)


def lut_record(mm, layout, bsize, z_order):
    b = mm.BlocksparseMatMul(layout, block_size=bsize, feature_axis=0, z_order=z_order)
    return dict(
        layout=np.asarray(layout, dtype=np.uint8), bsize=bsize, z_order=int(z_order),
        blocks=b.blocks, fprop_lut=b.fprop_lut, bprop_lut=b.bprop_lut, updat_lut=b.updat_lut,
        l2_lut=b.l2_lut, fprop_segments=b.fprop_segments, bprop_segments=b.bprop_segments,
        fprop_locks=b.fprop_locks, bprop_locks=b.bprop_locks, fprop_shared=b.fprop_shared,
        bprop_shared=b.bprop_shared, l2_shared=b.l2_shared)


def math_record(mm, layout, bsize, axis, N, seed):
    # the reference ctor only admits axis 1 for bsize 32/64 (blocksparse/matmul.py:84-89) although its
    # NumPy *_test functions are size-agnostic: build with axis 0, then flip the attribute.
    b = mm.BlocksparseMatMul(layout, block_size=bsize, feature_axis=0)
    b.axis = axis
    rng = np.random.RandomState(seed)
    f16 = lambda a: a.astype(np.float16).astype(np.float32)   # as test/blocksparse_matmul_test.py:313,345
    W = f16(rng.normal(0.0, 0.01, b.w_shape))
    X = f16(rng.normal(0.0, 0.1, b.i_shape(N)))
    E = f16(rng.normal(0.0, 0.1, b.o_shape(N)))
    Y = b.fprop_test(X, W)
    DX = b.bprop_test(E, W)
    DW = b.updat_test(X, E)
    # inputs are exactly representable in fp16, so they are stored as fp16 (lossless, 2x smaller);
    # outputs (float64 in the reference) are stored as float32.
    return dict(layout=np.asarray(layout, dtype=np.uint8), bsize=bsize, axis=axis, N=N, seed=seed,
                W=W.astype(np.float16), X=X.astype(np.float16), E=E.astype(np.float16),
                Y=Y.astype(np.float32), DX=DX.astype(np.float32), DW=DW.astype(np.float32))


def cfg0_inputs(w_shape, i_shape, o_shape, seed):
    """The exact inputs ``math_record`` draws (used by tests to regenerate cfg0 inputs)."""
    rng = np.random.RandomState(seed)
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    W = f16(rng.normal(0.0, 0.01, w_shape))
    X = f16(rng.normal(0.0, 0.1, i_shape))
    E = f16(rng.normal(0.0, 0.1, o_shape))
    return W, X, E


def gate_inputs(blocks, seed):
    """Per-block gates of the gated fixtures: a third exactly 0, the rest fp16-representable values in (0.25, 1.75)."""
    rng = np.random.RandomState(seed)
    g = rng.uniform(0.25, 1.75, blocks).astype(np.float16).astype(np.float32)
    g[rng.permutation(blocks)[: blocks // 3]] = 0.0
    return g


def main_gate():
    """gate.npz: the reference's gated fprop_test / bprop_test / updat_test(dw_gated=True) (feature axis 0 only there,
    blocksparse/matmul.py:367-373,391-397,412-418).  Inputs are regenerated by the tests (cfg0_inputs + gate_inputs)."""
    mm = import_reference_matmul()
    holes = ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    out = {}
    for name, lay, bs, N, seed in (("holes", holes, 32, 40, 31), ("holes", holes, 16, 24, 32), ("holes", holes, 8, 64, 33)):
        b = mm.BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
        W, X, E = cfg0_inputs(b.w_shape, b.i_shape(N), b.o_shape(N), seed)
        g = gate_inputs(b.blocks, seed)
        key = "%s/bs%d/" % (name, bs)
        out[key + "layout"] = np.asarray(lay, dtype=np.uint8)
        out[key + "meta"] = np.array([bs, N, seed], dtype=np.int64)
        out[key + "Y"] = b.fprop_test(X, W, gate=g).astype(np.float32)
        out[key + "DX"] = b.bprop_test(E, W, gate=g).astype(np.float32)
        out[key + "DW"] = b.updat_test(X, E, gate=g, dw_gated=True).astype(np.float32)
    path = os.path.join(HERE, "gate.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main_l2():
    """l2norm.npz: the reference's l2_normalize_test / l2_normalize_grad_test (blocksparse/matmul.py:421-445); inputs are
    regenerated by the tests with cfg0_inputs (W) and RandomState(seed + 1) (U)."""
    mm = import_reference_matmul()
    holes = ba_layout(16, 2, seed=3)
    holes[:, 5] = 0
    holes[7, :] = 0
    rng0 = np.random.default_rng(1234)
    rect = (rng0.random((6, 10)) < 0.3).astype(np.int32)
    rect[0, :] = 1
    rect[:, 0] = 1
    out = {}
    for name, lay, bs, seed in (("holes", holes, 32, 41), ("holes", holes, 8, 42), ("rect", rect, 16, 43)):
        b = mm.BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
        W, _, _ = cfg0_inputs(b.w_shape, (1, 1), (1, 1), seed)
        W[0, :, 0] = 0.0                                   # (only reaches the epsilon branch if block 0 is alone in its column)
        U = np.random.RandomState(seed + 1).normal(0.0, 1.0, b.w_shape).astype(np.float16).astype(np.float32)
        key = "%s/bs%d/" % (name, bs)
        out[key + "layout"] = np.asarray(lay, dtype=np.uint8)
        out[key + "meta"] = np.array([bs, seed], dtype=np.int64)
        out[key + "Y"] = b.l2_normalize_test(W.astype(np.float64)).astype(np.float32)
        out[key + "DW"] = b.l2_normalize_grad_test(W.astype(np.float64), U.astype(np.float64).copy()).astype(np.float32)
    path = os.path.join(HERE, "l2norm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    mm = import_reference_matmul()

    # ---- lookup tables -------------------------------------------------------------------
    np.random.seed(0)
    rand128 = np.random.randint(2, size=(128, 128))          # BASELINE.json configs[0]
    ba160 = ba_layout(160, 5, seed=1)                          # test/blocksparse_matmul_test.py:276-280
    holes = ba_layout(16, 2, seed=3)
    holes[:, 5] = 0                                            # an empty output column ...
    holes[7, :] = 0                                            # ... and an empty input row
    single = np.ones((1, 1), dtype=np.int32)
    rng = np.random.default_rng(1234)
    rect = (rng.random((6, 10)) < 0.3).astype(np.int32)        # CB != KB
    rect[0, :] = 1
    rect[:, 0] = 1
    luts = {}
    for name, lay, bs in (("rand128", rand128, 32), ("ba160", ba160, 32), ("ba160_bs8", ba160, 8),
                          ("holes", holes, 16), ("single", single, 32), ("rect", rect, 16)):
        for z in (True, False):
            rec = lut_record(mm, lay, bs, z)
            for k, v in rec.items():
                luts["%s/z%d/%s" % (name, int(z), k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "luts.npz"), **luts)

    # ---- math --------------------------------------------------------------------------------
    small = ba_layout(16, 2, seed=2)
    cases = {}
    i = 0
    for lay_name, lay in (("ba16", small), ("holes", holes), ("rect", rect), ("single", single)):
        for bs in (8, 16, 32):
            for axis in (0, 1):
                N = 24 if lay_name != "single" else 8
                rec = math_record(mm, lay, bs, axis, N, seed=100 + i)
                i += 1
                for k, v in rec.items():
                    cases["%s/bs%d/a%d/%s" % (lay_name, bs, axis, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "math.npz"), **cases)

    # BASELINE.json configs[0]: layout=random(128,128) bs 32 N=64 fp32, both axes.  To bound the file
    # size the inputs are NOT stored: tests regenerate them with ``cfg0_inputs`` below
    # (np.random.RandomState is a frozen legacy stream), and DW keeps every 64th weight block.
    cfg0 = {}
    for axis in (0, 1):
        rec = math_record(mm, rand128, 32, axis, 64, seed=7 + axis)
        cfg0["a%d/seed" % axis] = np.asarray(7 + axis)
        cfg0["a%d/Y" % axis] = rec["Y"]
        cfg0["a%d/DX" % axis] = rec["DX"]
        cfg0["a%d/DW_every64" % axis] = rec["DW"][::64]
    np.savez_compressed(os.path.join(HERE, "cfg0_rand128.npz"), **cfg0)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gate":
        main_gate()
    elif len(sys.argv) > 1 and sys.argv[1] == "l2":
        main_l2()
    else:
        main()
