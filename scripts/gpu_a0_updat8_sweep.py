"""bsize 8, feature axis 0, weight gradient at short minibatches: run once per build (BSMM_LIB=... built with -DU8P_ON=0 = without the pair kernel; the default build picks by its cost model): hipGraph replays, us"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us

tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
shapes = [("2560 dense", np.ones((320, 320), dtype=np.int32)), ("7680 11.4 % BA", P.ba_layout(960, 56, seed=1)), ("4096 10 %", P.random_layout(512, 512, 0.1, 1234)),
          ("20480 1.4 % BA", P.ba_layout(2560, 18, seed=1))]
for name, lay in shapes:
    b = BlocksparseMatMul(lay, block_size=8, feature_axis=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
    for N in (64, 128, 256, 512, 1024):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        b.updat(x, dy, dw=dw); k = _lib.last_kernel() & 255
        print("%-22s %s (%d blocks) N %4d: k%-2d updat %6.1f" % (tag, name, b.blocks, N, k, graph_us(lambda: b.updat(x, dy, dw=dw))), flush=True)
