"""feature axis 0, bsize 32, fprop / bprop at short minibatches (hipGraph replays, us): run once per library build
(BSMM_LIB=.../libbsmm_noxs0.so built with -DXS0_NMAX=0 = without the small-minibatch kernel of bsmm_xsmall0.h; -DXS0_NMAX=100000 = always)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us

tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
shapes = [("2560 dense", np.ones((80, 80), dtype=np.int32)), ("7680 11.7 % BA", P.ba_layout(240, 14, seed=1)), ("4096 20 %", P.random_layout(128, 128, 0.2, 1234)),
          ("20480 1.7 % BA", P.ba_layout(640, 5, seed=1))]
for name, lay in shapes:
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    for N in (64, 128, 256, 512, 1024, 2048):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        b.fprop(x, w); k = _lib.last_kernel() & 255
        print("%-22s %s (%d blocks) N %4d: k%-2d fprop %6.1f bprop %6.1f" % (tag, name, b.blocks, N, k, graph_us(lambda: b.fprop(x, w)), graph_us(lambda: b.bprop(dy, w))), flush=True)
