"""gated vs ungated fprop / bprop at the bench shape (bsize 32, bf16, N = 8192), both axes: plan kernel (default) and per-segment kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib

def timeit(fn, reps=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

import numpy as np
for bs, axis in ((32, 1), (32, 0), (16, 0), (16, 1)):
    lay = P.random_layout(128, 128, 0.2, 1234) if bs == 32 else P.random_layout(256, 256, 0.1, 1234)
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=axis)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    x = (torch.randn(b.i_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    dy = (torch.randn(b.o_shape(8192), device="cuda", generator=g) * 0.1).bfloat16()
    gate = torch.rand(b.blocks, device="cuda", generator=g) * 2 - 0.5
    gate[::7] = 0
    out = []
    mask = (torch.rand(b.blocks, device="cuda", generator=g) < 0.8).float()
    for name, gt, var, img in (("ungated", None, 0, True), ("gated, in-kernel", gate, 0, False), ("gated, weight images", gate, 0, True),
                               ("0/1 mask, in-kernel", mask, 0, False), ("0/1 mask, weight image", mask, 0, True)):
        _lib.set_kernel_variant(var)
        b.gate_images = img
        b.fprop(x, w, gate=gt); k = _lib.last_kernel()
        out.append("%s f %.1f b %.1f (k%d)" % (name, timeit(lambda: b.fprop(x, w, gate=gt)), timeit(lambda: b.bprop(dy, w, gate=gt)), k))
    _lib.set_kernel_variant(0)
    print("bsize %d axis %d: %s" % (bs, axis, " | ".join(out)), flush=True)
