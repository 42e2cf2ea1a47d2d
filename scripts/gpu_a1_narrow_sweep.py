"""feature axis 1, bsize 16 / 8, fprop / bprop at short minibatches (hipGraph replays, us); run once per library build (-DXSN_NMAX=0: without
xsmall_narrow_kernel; -DXSN_NMAX=100000: always)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us
tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
for bs in (16, 8):
    f = 32 // bs
    for name, lay in (("4096 10 %", P.random_layout(128 * f, 128 * f, 0.1, 1234)), ("2560 dense", np.ones((80 * f, 80 * f), dtype=np.int32)), ("20480 ~1.5 % BA", P.ba_layout(640 * f, 5 * f - 1, seed=1))):
        b = BlocksparseMatMul(lay, block_size=bs, feature_axis=1)
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
        for N in (64, 128, 256, 512, 1024, 2048):
            x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
            dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
            b.fprop(x, w); k = _lib.last_kernel() & 255
            print("%-20s bs %2d %s (%d blocks) N %4d: k%-2d fprop %6.1f bprop %6.1f" % (tag, bs, name, b.blocks, N, k, graph_us(lambda: b.fprop(x, w)), graph_us(lambda: b.bprop(dy, w))), flush=True)
