import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
for axis in (0, 1):
    lay = P.random_layout(128, 128, 0.2, 1234)
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=axis)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(64), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(64), device="cuda", generator=g) * 0.1).bfloat16()
    for _ in range(50):
        b.fprop(x, w); b.bprop(dy, w)
    torch.cuda.synchronize()
lay = np.ones((80, 80), dtype=np.int32)
b = BlocksparseMatMul(lay, block_size=32, feature_axis=0)
w = (torch.randn(b.w_shape, device="cuda") * 0.05).bfloat16()
x = (torch.randn(b.i_shape(64), device="cuda") * 0.1).bfloat16()
for _ in range(50):
    b.fprop(x, w); b.bprop(x, w)
torch.cuda.synchronize()
