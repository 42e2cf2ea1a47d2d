"""feature axis 0, weight gradient at short minibatches: the plan kernels (variant 0) against the per-block kernels (variant 2), hipGraph replays, us"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us   # noqa

shapes = [("2560 dense", np.ones((80, 80), dtype=np.int32), 32), ("7680 11.7 % BA", P.ba_layout(240, 14, seed=1), 32), ("4096 20 %", P.random_layout(128, 128, 0.2, 1234), 32),
          ("20480 1.7 % BA", P.ba_layout(640, 5, seed=1), 32),
          ("2560 dense", np.ones((160, 160), dtype=np.int32), 16), ("7680 11.5 % BA", P.ba_layout(480, 28, seed=1), 16), ("4096 10 %", P.random_layout(256, 256, 0.1, 1234), 16),
          ("20480 1.5 % BA", P.ba_layout(1280, 9, seed=1), 16)]
for name, lay, bs in shapes:
    b = BlocksparseMatMul(lay, block_size=bs, feature_axis=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    for N in (64, 128, 256, 512, 1024, 2048):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        row = []
        for var in (0, 2):
            _lib.set_kernel_variant(var)
            b.updat(x, dy, dw=dw); k = _lib.last_kernel() & 255
            row.append("v%d k%d %.1f" % (var, k, graph_us(lambda: b.updat(x, dy, dw=dw))))
        _lib.set_kernel_variant(0)
        print("bs %d %s (%d blocks) N %d: %s" % (bs, name, b.blocks, N, " | ".join(row)), flush=True)
