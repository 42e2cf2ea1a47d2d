"""Medium-minibatch kernel (bsmm_xmid.h) in a build that forces it (-DBSMM_MID_MODE=1, BSMM_LIB=...): bit identity with the plan kernels
(BSMM_FLAG_FORCE_PLAN: flow / staged, same summation order) on a set of shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
cases = [("tiny 4x4 N=128", P.random_layout(4, 4, 0.6, seed=1), 128, torch.bfloat16),
         ("40x24 N=1000 (ragged rows)", P.random_layout(40, 24, 0.3, seed=2), 1000, torch.bfloat16),
         ("33x35 N=520", P.random_layout(33, 35, 0.25, seed=3), 520, torch.float16),
         ("128x128 20% N=512", P.random_layout(128, 128, 0.2, seed=1234), 512, torch.bfloat16),
         ("128x128 55% N=300 (lists > 64 entries)", P.random_layout(128, 128, 0.55, seed=5), 300, torch.bfloat16),
         ("300x16 5% N=640", P.random_layout(300, 16, 0.05, seed=6), 640, torch.bfloat16),
         ("256x256 5% N=512", P.random_layout(256, 256, 0.05, seed=1234), 512, torch.bfloat16),
         ("15x33 eye (columns without entries), N=70", np.eye(15, 33, dtype=np.int32), 70, torch.float16)]
ok = True
for name, lay, N, td in cases:
    b = BlocksparseMatMul(lay, block_size=32, feature_axis=1)
    g = torch.Generator(device="cuda").manual_seed(7)
    w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).to(td)
    x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).to(td)
    dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).to(td)
    y1 = b.fprop(x, w); k1 = lib.last_kernel()
    d1 = b.bprop(dy, w); k1b = lib.last_kernel()
    torch.cuda.synchronize()
    lib.set_kernel_variant(3)
    y2 = b.fprop(x, w); k2 = lib.last_kernel()
    d2 = b.bprop(dy, w)
    torch.cuda.synchronize()
    lib.set_kernel_variant(0)
    e1, e2 = torch.equal(y1, y2), torch.equal(d1, d2)
    print("%-45s kernels %d/%d vs %d  fprop identical %s  bprop identical %s  (max |diff| %.3e / %.3e)" %
          (name, k1, k1b, k2, e1, e2, (y1.float() - y2.float()).abs().max().item(), (d1.float() - d2.float()).abs().max().item()), flush=True)
    ok = ok and e1 and e2 and k1 == lib.K_XPROP_MID and k1b == lib.K_XPROP_MID
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
