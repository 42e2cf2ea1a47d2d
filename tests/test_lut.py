"""Product LUT builder (blocksparse_amd/lut.py) == reference tables (golden) == oracle restatement."""
import os

import numpy as np
import pytest

from blocksparse_amd import lut as L
from oracle import bsmm_oracle as orc


@pytest.fixture(scope="module")
def luts(golden_dir):
    return np.load(os.path.join(golden_dir, "luts.npz"))


@pytest.mark.parametrize("name", ["rand128", "ba160", "ba160_bs8", "holes", "single", "rect"])
@pytest.mark.parametrize("z", [1, 0])
def test_builder_matches_reference(luts, name, z):
    g = lambda k: luts["%s/z%d/%s" % (name, z, k)]
    t = L.build_tables(g("layout"), z_order=bool(z))
    assert t["blocks"] == int(g("blocks"))
    np.testing.assert_array_equal(t["updat_lut"], g("updat_lut"))
    for side in ("fprop", "bprop"):
        np.testing.assert_array_equal(t[side]["lut"], g(side + "_lut"), err_msg=side)
        assert t[side]["segments"] == int(g(side + "_segments"))
        assert t[side]["locks"] == int(g(side + "_locks"))
        assert t[side]["shared"] == int(g(side + "_shared"))
    np.testing.assert_array_equal(t["fprop"]["l2_lut"], g("l2_lut"))
    assert t["fprop"]["l2_shared"] == int(g("l2_shared"))


def test_z_order_matches_oracle_scalar():
    rng = np.random.default_rng(0)
    xs = rng.integers(0, 70000, size=200)
    ys = rng.integers(0, 70000, size=200)
    vec = L.z_order_2d(xs, ys)
    for x, y, v in zip(xs, ys, vec):
        assert int(v) == orc.z_order_2d(int(x), int(y)) == L.z_order_2d(int(x), int(y))


@pytest.mark.parametrize("seed", range(6))
def test_builder_matches_oracle_random_layouts(seed):
    rng = np.random.default_rng(seed)
    CB, KB = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    lay = rng.random((CB, KB)) < rng.uniform(0.05, 0.9)
    lay[rng.integers(0, CB), rng.integers(0, KB)] = True
    # make it skewed so that segmentation + locks trigger sometimes
    lay[:, 0] = True
    for z in (True, False):
        t = L.build_tables(lay, z_order=z)
        o = orc.build_layout_luts(lay, 8, z)
        np.testing.assert_array_equal(t["updat_lut"], o["updat_lut"])
        np.testing.assert_array_equal(t["fprop"]["lut"], o["fprop_lut"])
        np.testing.assert_array_equal(t["bprop"]["lut"], o["bprop_lut"])
        assert t["fprop"]["locks"] == o["fprop_locks"] and t["bprop"]["locks"] == o["bprop_locks"]
        assert t["fprop"]["cols"] == o["fprop_list"] and t["bprop"]["cols"] == o["bprop_list"]


def test_unsegmented_has_no_locks(luts):
    lay = luts["ba160/z1/layout"]
    t = L.build_tables(lay, segmented=False)
    assert t["fprop"]["locks"] == 0 and t["bprop"]["locks"] == 0
    assert t["fprop"]["segments"] == lay.shape[1] and t["bprop"]["segments"] == lay.shape[0]
    # same entries, same order, just one header per output block
    s = L.build_tables(lay, segmented=True)
    np.testing.assert_array_equal(t["fprop"]["lut"][4 * t["fprop"]["segments"]:],
                                  s["fprop"]["lut"][4 * s["fprop"]["segments"]:])


def test_big_layout_builds_fast():
    import time
    rng = np.random.default_rng(1)
    lay = rng.random((1024, 1024)) < 0.2          # ~210k blocks
    t0 = time.time()
    t = L.build_tables(lay)
    assert time.time() - t0 < 5.0
    assert t["blocks"] == int(lay.sum())


def test_double_tables_for_gated_calls():
    """lut.double_tables (round 6: gated calls on the ungated kernels over hi / lo weight images): every entry (c, w) is followed by
    (c, w + blocks), headers keep their order, lengths and offsets double, and the doubled table is a valid xprop table for every plan builder."""
    import ctypes
    from blocksparse_amd import _lib as lib
    from blocksparse_amd.matmul import _host_plan
    import _parity
    lay = _parity.random_layout(24, 40, 0.3, seed=5)
    for segmented in (False, True):
        t = L.build_tables(lay, z_order=True, segmented=segmented)
        d = L.double_tables(t)
        B = t["blocks"]
        assert d["blocks"] == 2 * B
        for side, n_out in (("fprop", t["KB"]), ("bprop", t["CB"])):
            a, b = t[side], d[side]
            S = a["segments"]
            assert b["segments"] == S and b["locks"] == a["locks"] and b["shared"] == 2 * a["shared"]
            la, lb = np.asarray(a["lut"]), np.asarray(b["lut"])
            assert lb.size == 4 * S + 4 * B
            for s_ in range(S):
                off, cnt, ob, lock = la[4 * s_:4 * s_ + 4]
                off2, cnt2, ob2, lock2 = lb[4 * s_:4 * s_ + 4]
                assert (cnt2, ob2, lock2) == (2 * cnt, ob, lock)
                ea = la[2 * off:2 * (off + cnt)].reshape(-1, 2)
                eb = lb[2 * off2:2 * (off2 + cnt2)].reshape(-1, 2, 2)
                assert np.array_equal(eb[:, 0, :], ea) and np.array_equal(eb[:, 1, 0], ea[:, 0]) and np.array_equal(eb[:, 1, 1], ea[:, 1] + B)
            if not segmented:
                for bs, axis, opt in ((32, 1, lib.PLAN_XCOL_FLOW), (32, 0, 0), (16, 0, 0), (16, 1, 0)):
                    assert _host_plan(b["lut"], S, 2 * B, n_out, bs, lib.BF16, axis, opt) is not None
