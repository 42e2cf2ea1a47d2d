"""Host-side checks of the bsize-8 'BSS8' plans (blocksparse_amd/csrc/bsmm_plan.h): the 8x8 blocks are grouped into the
32x32 super-blocks of the block grid and the bsize-32 matrix-core kernels run on that super layout.  Here the plan is
interpreted in numpy exactly the way the expand8 / gather8 kernels do and compared with the dense matrices of the
bsize-8 layout.  No GPU needed (the builders are pure host code inside the library)."""
import numpy as np
import pytest

from blocksparse_amd import _lib
from blocksparse_amd import lut as _lut
from blocksparse_amd.matmul import _host_plan, _host_updat_plan
from tests import _parity as P

S8_MAGIC, XC_MAGIC, UP_MAGIC, X2_MAGIC, U2_MAGIC = 0x42535338, 0x42535843, 0x42535550, 0x42535832, 0x42535532


def _dense(layout, W, bs):
    CB, KB = layout.shape
    D = np.zeros((CB * bs, KB * bs), dtype=W.dtype)
    t = _lut.build_tables(layout, z_order=True, segmented=False)
    for w, (c, k) in enumerate(t["updat_lut"].tolist()):
        D[c * bs:(c + 1) * bs, k * bs:(k + 1) * bs] = W[w]
    return D, t


def _parts(plan):
    assert plan[0] == S8_MAGIC and plan[1] == 1
    ns = int(plan[2])
    sub = plan[plan[3]:plan[3] + 16 * ns].reshape(ns, 4, 4)
    lut32 = plan[plan[4]:plan[4] + 2 * ns].reshape(ns, 2)
    nested = plan[plan[5]:plan[6]]
    assert plan[5] % 4 == 0 and len(plan) == plan[6]
    return ns, sub, lut32, nested


@pytest.mark.parametrize("shape,density", [((8, 12), 0.3), ((40, 40), 0.1), ((4, 4), 1.0), ((16, 8), 0.02)])
def test_xprop_super_plans_reproduce_the_dense_matrix(shape, density):
    layout = P.random_layout(shape[0], shape[1], density, seed=5)
    rng = np.random.default_rng(0)
    t = _lut.build_tables(layout, z_order=True, segmented=False)
    B = t["blocks"]
    W = rng.normal(size=(B, 8, 8)).astype(np.float32)
    D, _ = _dense(layout, W, 8)
    for side, n_out, transposed in (("fprop", t["KB"], False), ("bprop", t["CB"], True)):
        plan = _host_plan(t[side]["lut"], t[side]["segments"], B, n_out, 8, _lib.BF16, 1)
        assert plan is not None
        ns, sub, lut32, nested = _parts(plan)
        assert nested[0] == XC_MAGIC and nested[8] == n_out // 4
        # every 8x8 block sits in exactly one sub slot
        ids = sub[sub >= 0]
        assert sorted(ids.tolist()) == list(range(B))
        # expand the way expand8_kernel does: Wsel[s][out32][in32]; fprop blocks are stored [in][out], bprop [out][in]
        M = D.T if transposed else D          # M[in feature][out feature] of this pass
        got = np.zeros_like(M)
        for s in range(ns):
            in32, out32 = lut32[s]
            blk = np.zeros((32, 32), dtype=np.float32)          # [out][in]
            for a in range(4):
                for b in range(4):
                    w = sub[s, a, b]
                    if w >= 0:
                        src = W[w].T if side == "fprop" else W[w]      # -> [out][in]
                        blk[8 * b:8 * b + 8, 8 * a:8 * a + 8] = src
            got[in32 * 32:(in32 + 1) * 32, out32 * 32:(out32 + 1) * 32] = blk.T
        np.testing.assert_array_equal(got, M)
        # the nested plan lists every super-block exactly once
        ng, off_g, off_p, off_w = int(nested[3]), int(nested[5]), int(nested[6]), int(nested[7])
        wt = nested[off_w:]
        assert sorted(wt[wt >= 0].tolist()) == list(range(ns))


@pytest.mark.parametrize("shape,density", [((8, 12), 0.3), ((40, 40), 0.1), ((4, 4), 1.0)])
def test_updat_super_plan_covers_every_block_once(shape, density):
    layout = P.random_layout(shape[0], shape[1], density, seed=9)
    t = _lut.build_tables(layout, z_order=True, segmented=False)
    B = t["blocks"]
    plan = _host_updat_plan(t["updat_lut"], B, t["CB"], t["KB"], 8, _lib.BF16, 1)
    ns, sub, lut32, nested = _parts(plan)
    assert nested[0] == U2_MAGIC and nested[5] == ns       # round 3: the streaming kernel's plan of the super layout
    for w, (c, k) in enumerate(t["updat_lut"].tolist()):
        s = np.nonzero((lut32[:, 0] == c // 4) & (lut32[:, 1] == k // 4))[0]
        assert len(s) == 1 and sub[s[0], c % 4, k % 4] == w
    assert (sub >= 0).sum() == B
    assert len({tuple(r) for r in lut32.tolist()}) == ns


def test_no_super_plan_when_the_grid_is_not_a_multiple_of_four_or_fp32():
    layout = P.random_layout(7, 9, 0.5, seed=2)
    t = _lut.build_tables(layout, z_order=True, segmented=False)
    assert _host_plan(t["fprop"]["lut"], t["fprop"]["segments"], t["blocks"], t["KB"], 8, _lib.BF16, 1) is None
    assert _host_updat_plan(t["updat_lut"], t["blocks"], t["CB"], t["KB"], 8, _lib.BF16, 0) is None
    layout = P.random_layout(8, 8, 0.5, seed=2)
    t = _lut.build_tables(layout, z_order=True, segmented=False)
    assert _host_plan(t["fprop"]["lut"], t["fprop"]["segments"], t["blocks"], t["KB"], 8, _lib.F32, 1) is None
