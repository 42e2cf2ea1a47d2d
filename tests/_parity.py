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


# ---- the ONE per-block criterion -------------------------------------------------------------------
# A result is compared with the float64 oracle rounded ONCE to the storage type.  Three statements, all of which must hold:
#   (1) every element: got and the rounded oracle value are the SAME or ADJACENT representable values of the storage type
#       (the kernels accumulate in fp32 and round once, the oracle accumulates in float64 and rounds once: the two sums differ
#       by fp32 noise, which can flip a rounding but cannot move it two steps) -- OR the two differ by less than
#       ELEM_FLOOR x the block's rms (elements that cancelled to near zero, where fp32 noise spans many of their tiny ulps);
#   (2) every block (256 .. 1024 elements, or whatever group the caller reshapes to): L2-relative error <= BLOCK_FACTOR x bar.
#       A block of 256 bf16 values that all sit one fp32-noise away from a rounding boundary reaches 0.75 - 1.2e-3 through
#       1-ulp flips alone (Monte-Carlo in VERDICT r5), so 1 x bar per block is a coin toss; 4 x is what the suite used for DW
#       blocks since round 1;
#   (3) the whole tensor: L2-relative error <= bar (the north-star statement).
# fp32 storage has no statement (1): the bar there (2e-6) is already a few fp32 ulps of the L2 norm.
ELEM_FLOOR = 1e-4
BLOCK_FACTOR = 4.0


def _ordinal(a, dtype):
    """Monotonic integer index of the representable values of a 16-bit storage type (adjacent values differ by 1)."""
    if dtype == "bf16":
        bits = (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.int64)
    elif dtype == "f16":
        bits = np.ascontiguousarray(a).astype(np.float16).view(np.uint16).astype(np.int64)
    else:
        raise ValueError(dtype)
    mag = bits & 0x7FFF
    return np.where(bits & 0x8000, -mag, mag)


def block_report(got, want64, dtype, groups):
    """`got`: the device result (float32 array holding storage-type values); `want64`: the UNROUNDED float64 oracle; `groups`: the
    leading dimension to reshape to (one row = one block).  Returns a dict of the three statements' worst cases."""
    got = np.asarray(got)
    want_r = round_to_storage(want64, dtype)
    g = got.astype(np.float64).reshape(groups, -1)
    r = want_r.astype(np.float64).reshape(groups, -1)
    d = np.abs(g - r)
    den = np.sqrt((r ** 2).sum(axis=1))
    num = np.sqrt((d ** 2).sum(axis=1))
    live = den > 0
    rep = {"tensor_l2": float(np.sqrt((d ** 2).sum()) / max(np.sqrt((r ** 2).sum()), 1e-30)),
           "block_l2": float((num[live] / den[live]).max()) if live.any() else 0.0,
           "worst_block": int(np.argmax(np.where(live, num / np.maximum(den, 1e-30), 0.0))),
           "dead_nonzero": int((np.abs(g[~live]).sum(axis=1) != 0).sum()),
           "finite": bool(np.isfinite(got).all()), "elem_bad": 0, "elem_flips": 0}
    if dtype in ("bf16", "f16"):
        steps = np.abs(_ordinal(got.astype(np.float32), dtype) - _ordinal(want_r.astype(np.float32), dtype)).reshape(groups, -1)
        rms = den / np.sqrt(r.shape[1])
        bad = (steps > 1) & (d > ELEM_FLOOR * rms[:, None])
        rep["elem_bad"] = int(bad.sum())
        rep["elem_flips"] = int((steps == 1).sum())
    return rep


def assert_blocks(got, want64, dtype, groups, ctx=""):
    rep = block_report(got, want64, dtype, groups)
    bar = L2_BAR[dtype]
    assert rep["finite"], (ctx, "non-finite values")
    assert rep["dead_nonzero"] == 0, (ctx, "blocks whose oracle is all zero are not", rep)
    assert rep["elem_bad"] == 0, (ctx, "elements more than one storage-type step from the rounded oracle", rep)
    assert rep["block_l2"] <= BLOCK_FACTOR * bar, (ctx, "worst block", rep)
    assert rep["tensor_l2"] <= bar, (ctx, "tensor", rep)
    return rep


def round_to_storage(a, dtype):
    return orc.round_to(np.asarray(a), dtype)


def act_blocks(a, axis, N, feat_blocks, bs):
    """An activation-shaped array as (feature block, everything else): the grouping of the per-block-column statement."""
    a = np.asarray(a)
    return (a.reshape(N, feat_blocks, bs).transpose(1, 0, 2) if axis else a.reshape(feat_blocks, bs, N)).reshape(feat_blocks, -1)


def gen(torch, seed, device="cuda"):
    """A seeded generator: no test of this suite draws from the global RNG state."""
    return torch.Generator(device=device).manual_seed(int(seed))
