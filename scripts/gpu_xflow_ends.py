"""When do the workgroups (and their waves) of the flow kernel finish their units?  Wall-clock stamps of a -DX4_ENDSTAMPS build
(BSMM_LIB=.../libbsmm_x4ends.so), 4096^2 bsize 32 bf16 feature axis 1 N = 8192, density argv[1] (default 20), fprop and bprop."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _parity as P
from blocksparse_amd import BlocksparseMatMul, _lib as lib
L = lib.load()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(os.environ.get("N", "8192"))
b = BlocksparseMatMul(P.random_layout(128, 128, d / 100.0, 1234), block_size=32, feature_axis=1)
g = torch.Generator(device="cuda").manual_seed(1)
w = (torch.randn(b.w_shape, device="cuda", generator=g) * 0.05).bfloat16()
x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
for name, fn in (("fprop", lambda: b.fprop(x, w)), ("bprop", lambda: b.bprop(dy, w))):
    for rep in range(2):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        assert lib.last_kernel() == lib.K_XCOL32_FLOW
        buf = np.zeros(512 * 16 * 8, dtype=np.uint64)
        assert L.bsmm_debug_x4_ends_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        t = buf.reshape(512, 16, 8).astype(np.float64)
        n = int((t[:, 0, 0] > 0).sum())
        t = t[:n]
        t0 = t[:, :, 0].min()
        units = int(t[:, 0, 7].max())
        us = lambda a: (a - t0) / 100.0
        print("d%d N %d %s run %d: %d workgroups, units per workgroup %s" % (d, N, name, rep, n, np.unique(t[:, 0, 7]).astype(int).tolist()))
        print("  %-34s %8s %8s %8s %8s" % ("us from the first start", "min", "median", "p90", "max"))
        for u in range(min(units, 6)):
            wg_end = us(t[:, :, 1 + u]).max(axis=1)        # the unit is stored when its last wave has stored
            wg_first = us(t[:, :, 1 + u]).min(axis=1)
            print("  unit %d stored (last wave)          %8.1f %8.1f %8.1f %8.1f" % (u, wg_end.min(), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max()))
            print("  unit %d stored (first wave)         %8.1f %8.1f %8.1f %8.1f" % (u, wg_first.min(), np.median(wg_first), np.percentile(wg_first, 90), wg_first.max()))
        last = us(t[:, :, units]).max(axis=1)
        print("  per XCD: end median / max:", "  ".join("%d: %.1f / %.1f" % (xc, np.median(last[np.arange(n) % 8 == xc]), last[np.arange(n) % 8 == xc].max()) for xc in range(8)))
