"""Sparse-weight checkpoint format and dense <-> sparse converters (CPU only)."""
import numpy as np
import pytest
import torch

from blocksparse_amd import BlocksparseMatMul, checkpoint as ck
from oracle import bsmm_oracle as orc


def _layout(seed=0):
    rng = np.random.default_rng(seed)
    lay = rng.random((7, 11)) < 0.4
    lay[0, 0] = True
    return lay


@pytest.mark.parametrize("z", [True, False])
def test_dense_roundtrip_matches_oracle(z):
    lay = _layout()
    bs = 8
    t = orc.build_layout_luts(lay, bs, z)
    W = np.random.default_rng(1).normal(size=(t["blocks"], bs, bs)).astype(np.float32)
    Wd = ck.to_dense(lay, W, z_order=z)
    np.testing.assert_array_equal(Wd, orc.to_dense(t, W).astype(np.float32))
    np.testing.assert_array_equal(ck.from_dense(lay, Wd, bs, z_order=z), W)
    # outside the layout the dense matrix is exactly zero
    mask = np.kron(lay, np.ones((bs, bs))).astype(bool)
    assert not Wd[~mask].any()


def test_renumber_between_z_order_flags():
    lay = _layout(3)
    bs = 16
    W_z = np.random.default_rng(2).normal(size=(int(lay.sum()), bs, bs)).astype(np.float32)
    W_plain = ck.renumber(W_z, lay, True, False)
    np.testing.assert_array_equal(ck.to_dense(lay, W_z, True), ck.to_dense(lay, W_plain, False))
    np.testing.assert_array_equal(ck.renumber(W_plain, lay, False, True), W_z)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_save_load_roundtrip(tmp_path, dtype):
    lay = _layout(5)
    b = BlocksparseMatMul(lay, block_size=16, feature_axis=1, z_order=False)
    W = torch.randn(b.w_shape, generator=torch.Generator().manual_seed(7)).to(dtype)
    path = str(tmp_path / "w.npz")
    ck.save(path, b, W)
    b2, W2 = ck.load(path, device="cpu")
    assert (b2.bsize, b2.axis, b2.z_order) == (16, 1, False)
    np.testing.assert_array_equal(b2.layout, b.layout)
    np.testing.assert_array_equal(b2.fprop_lut, b.fprop_lut)
    assert W2.dtype == dtype and torch.equal(W2, W)
    b3, W3 = ck.load(path)          # NumPy form
    assert W3.shape == b.w_shape


def test_shape_mismatch_is_rejected(tmp_path):
    b = BlocksparseMatMul(_layout(), block_size=8)
    with pytest.raises(ValueError):
        ck.save(str(tmp_path / "x.npz"), b, np.zeros((3, 8, 8), dtype=np.float32))
    with pytest.raises(ValueError):
        ck.from_dense(_layout(), np.zeros((5, 5)), 8)
