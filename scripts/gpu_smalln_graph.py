"""feature axis 1, bsize 32, bf16: fprop / bprop at short minibatches as hipGraph replays (us); run once per library build (BSMM_LIB)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib
from gpu_ref_bench_shapes import graph_us
tag = os.path.basename(os.environ.get("BSMM_LIB", "default"))
for name, lay in (("4096 20 %", P.random_layout(128, 128, 0.2, 1234)), ("8192 5 %", P.random_layout(256, 256, 0.05, 1234)), ("2560 dense", np.ones((80, 80), dtype=np.int32))):
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
    for N in (64, 128, 256, 512):
        x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
        b.fprop(x, w); k = _lib.last_kernel() & 255
        print("%-16s %s N %4d: k%-2d fprop %6.1f bprop %6.1f" % (tag, name, N, k, graph_us(lambda: b.fprop(x, w)), graph_us(lambda: b.bprop(dy, w))), flush=True)
