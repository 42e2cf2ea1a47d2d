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
BS = int(os.environ.get("BS", "32"))
f = 32 // BS
shapes = [("2560 dense", np.ones((80 * f, 80 * f), dtype=np.int32)), ("7680 ~11.5 % BA", P.ba_layout(240 * f, 14 * f, seed=1)),
          ("4096 %d %%" % (20 if BS == 32 else 10), P.random_layout(128 * f, 128 * f, 0.2 if BS == 32 else 0.1, 1234)), ("20480 ~1.5 % BA", P.ba_layout(640 * f, 5 * f - (f > 1), seed=1))]
for name, lay in shapes:
    b = BlocksparseMatMul(lay, block_size=BS, feature_axis=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    for N in (64, 128, 256, 512, 1024, 2048):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        b.fprop(x, w); k = _lib.last_kernel() & 255
        print("%-22s bs %d %s (%d blocks) N %4d: k%-2d fprop %6.1f bprop %6.1f" % (tag, BS, name, b.blocks, N, k, graph_us(lambda: b.fprop(x, w)), graph_us(lambda: b.bprop(dy, w))), flush=True)
