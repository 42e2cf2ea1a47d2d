"""SparseProj (gather / scatter / scatter_add / scatter_mul with gradients): table construction as the reference class
(blocksparse/matmul.py:845-880), device ops against the NumPy statements, gradients against torch indexing."""
import os
import sys

import numpy as np
import pytest

import _parity as P

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_tables():
    from blocksparse_amd import SparseProj
    sp = SparseProj(1024, proj_stride=4, block_size=32)
    assert sp.nproj == 256 and np.array_equal(sp.gather_lut, np.arange(0, 1024, 4))
    sp = SparseProj(1000, proj_stride=3, block_size=32)                 # trimmed to a multiple of block_size
    assert sp.nproj == (1000 // 3 // 32) * 32 and sp.gather_lut[-1] == (sp.nproj - 1) * 3
    np.random.seed(0)
    sp = SparseProj(64, nproj=10)
    assert sp.nproj == 10 and np.all(np.diff(sp.gather_lut) > 0)
    assert np.array_equal(np.nonzero(sp.scatter_lut >= 0)[0], sp.gather_lut)
    assert np.array_equal(sp.scatter_lut[sp.gather_lut], np.arange(10))
    sp2 = SparseProj.__new__(SparseProj)
    sp2.__setstate__(sp.__getstate__())
    assert np.array_equal(sp2.gather_lut, sp.gather_lut) and sp2.nhidden == 64
    with pytest.raises(ValueError):
        SparseProj(64)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_ops_and_gradients(dtype):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import __graft_entry__ as g
    g.build()
    from blocksparse_amd import SparseProj
    td = getattr(torch, dtype)
    np.random.seed(1)
    for nhidden, nproj, N in ((96, 17, 40), (512, 128, 1000), (33, 33, 7)):
        sp = SparseProj(nhidden, nproj=nproj)
        idx = torch.from_numpy(sp.gather_lut.astype(np.int64)).cuda()
        x = torch.randn(nhidden, N, device="cuda", generator=P.gen(torch, nhidden)).to(td)
        y = torch.randn(nproj, N, device="cuda", generator=P.gen(torch, nproj + 1000)).to(td)
        assert torch.equal(sp.gather(x), x[idx])
        z = torch.zeros_like(x); z[idx] = y
        assert torch.equal(sp.scatter(y), z)
        ref = x.clone(); ref[idx] = (x[idx].float() + y.float()).to(td)
        assert torch.equal(sp.scatter_add(x, y), ref)
        ref = x.clone(); ref[idx] = (x[idx].float() * y.float()).to(td)
        assert torch.equal(sp.scatter_mul(x, y), ref)
        if dtype == "float32":                                               # the class's NumPy statements agree with the device
            xn, yn = x.cpu().numpy(), y.cpu().numpy()
            assert np.array_equal(sp.gather_test(xn), sp.gather(x).cpu().numpy())
            assert np.array_equal(sp.scatter_test(yn), sp.scatter(y).cpu().numpy())
            assert np.array_equal(sp.scatter_add_test(xn, yn), sp.scatter_add(x, y).cpu().numpy())
            assert np.array_equal(sp.scatter_mul_test(xn, yn), sp.scatter_mul(x, y).cpu().numpy())
        # gradients (fp32 only for exact comparison)
        if dtype == "float32":
            xa = x.clone().requires_grad_(True); ya = y.clone().requires_grad_(True)
            e = torch.randn(nhidden, N, device="cuda", generator=P.gen(torch, N + 2000))
            sp.scatter_mul(sp.scatter_add(xa, ya), ya).backward(e)
            xb = x.clone().requires_grad_(True); yb = y.clone().requires_grad_(True)
            t = xb.clone(); t = t.index_add(0, idx, yb)
            m = torch.ones_like(t).index_copy(0, idx, yb)
            (t * m).backward(e)
            assert torch.allclose(xa.grad, xb.grad, rtol=1e-6, atol=1e-6) and torch.allclose(ya.grad, yb.grad, rtol=1e-5, atol=1e-5)
            xg = x.clone().requires_grad_(True)
            sp.scatter(sp.gather(xg)).backward(e)
            mask = torch.zeros(nhidden, 1, device="cuda"); mask[idx] = 1
            assert torch.equal(xg.grad, e * mask)
    with pytest.raises(ValueError):
        sp.gather(torch.zeros(5, 4, device="cuda"))
