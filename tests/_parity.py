"""Shared helpers for the GPU parity tests (HIP path through the C ABI vs. oracle/bsmm_oracle.py)."""
import numpy as np

from oracle import bsmm_oracle as orc

TORCH_DT = {"f32": "float32", "f16": "float16", "bf16": "bfloat16"}
# L2-relative bars (north_star: <=1e-3 rel-err; fp32 far tighter).  16-bit results are compared with the
# oracle evaluated on the SAME rounded inputs in float64 and rounded ONCE to the storage type.
L2_BAR = {"f32": 2e-6, "f16": 1e-3, "bf16": 1e-3}
MAX_BAR = {"f32": 2e-5, "f16": 1e-2, "bf16": 6e-2}   # max|diff| / mean|ref|  (one 16-bit ulp flip on the largest element)


def ba_layout(n, m, seed):
    """Barabasi-Albert adjacency + I with a dense m x m corner (test/blocksparse_matmul_test.py:276-280),
    generated without networkx so that it also runs on the GPU box."""
    rng = np.random.RandomState(seed)
    lay = np.eye(n, dtype=np.int32)
    targets = list(range(m))
    repeated = []
    for src in range(m, n):
        for t in set(targets):
            lay[src, t] = lay[t, src] = 1
        repeated.extend(targets)
        repeated.extend([src] * m)
        targets = []
        while len(set(targets)) < m:
            targets.append(repeated[rng.randint(len(repeated))])
        targets = list(set(targets))[:m]
    lay[0:m, 0:m] = 1
    return lay


def random_layout(CB, KB, density, seed):
    rng = np.random.default_rng(seed)
    lay = (rng.random((CB, KB)) < density).astype(np.int32)
    for r in np.nonzero(lay.sum(axis=1) == 0)[0]:        # >= 1 block per row
        lay[r, rng.integers(0, KB)] = 1
    for c in np.nonzero(lay.sum(axis=0) == 0)[0]:        # >= 1 block per column
        lay[rng.integers(0, CB), c] = 1
    return lay


def make_inputs(w_shape, i_shape, o_shape, dtype, seed):
    """W ~ N(0, .01), X, E ~ N(0, .1), pre-rounded through fp16 (test/blocksparse_matmul_test.py:313,345-346)
    and then to the storage dtype so every path sees exactly representable inputs."""
    rng = np.random.RandomState(seed)
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    W = orc.round_to(f16(rng.normal(0.0, 0.01, w_shape)), dtype)
    X = orc.round_to(f16(rng.normal(0.0, 0.1, i_shape)), dtype)
    E = orc.round_to(f16(rng.normal(0.0, 0.1, o_shape)), dtype)
    return W, X, E


def errors(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(got - ref)
    l2 = np.sqrt((d ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30)
    mx = d.max() / max(np.abs(ref).mean(), 1e-30)
    return l2, mx


def to_dev(a, dtype, torch):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").to(getattr(torch, TORCH_DT[dtype]))


def to_host(t):
    return t.detach().float().cpu().numpy()


def run_case(torch, BlocksparseMatMul, layout, bs, axis, dtype, N, seed=0, segmented=False, passes=("Y", "DX", "DW"),
             fast_oracle=False):
    """Run fprop/bprop/updat on the GPU through the product API and return {pass: (l2, max)} vs the oracle."""
    bsmm = BlocksparseMatMul(layout, block_size=bs, feature_axis=axis, segmented=segmented)
    t = orc.build_layout_luts(layout, bs)
    W, X, E = make_inputs(bsmm.w_shape, bsmm.i_shape(N), bsmm.o_shape(N), dtype, seed)
    w, x, e = (to_dev(a, dtype, torch) for a in (W, X, E))
    out = {}
    if "Y" in passes:
        ref = (orc.fprop_fast(t, X, W, axis, np.float64) if fast_oracle else orc.fprop(t, X, W, axis))
        out["Y"] = errors(to_host(bsmm.fprop(x, w)), orc.round_to(ref, dtype))
    if "DX" in passes:
        ref = (orc.bprop_fast(t, E, W, axis, np.float64) if fast_oracle else orc.bprop(t, E, W, axis))
        out["DX"] = errors(to_host(bsmm.bprop(e, w)), orc.round_to(ref, dtype))
    if "DW" in passes:
        ref = (orc.updat_fast(t, X, E, axis, np.float64) if fast_oracle else orc.updat(t, X, E, axis))
        out["DW"] = errors(to_host(bsmm.updat(x, e)), orc.round_to(ref, dtype))
    return out
