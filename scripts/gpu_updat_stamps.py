"""When do the workgroups of the streaming weight-gradient kernel finish?  Wall-clock stamps of a -DU2_STAMPS build (BSMM_LIB=.../libbsmm_u2stamps.so),
4096^2 bsize 32 bf16 feature axis 1 N = 8192 at the density in argv[1] (default 20): per XCD (= blockIdx % 8) and over the chip, us from the first start."""
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
x = (torch.randn(b.i_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
dy = (torch.randn(b.o_shape(N), device="cuda", generator=g) * 0.1).bfloat16()
dw = torch.empty(b.w_shape, dtype=torch.bfloat16, device="cuda")
for rep in range(3):
    for _ in range(20): b.updat(x, dy, dw=dw)
    torch.cuda.synchronize()
    assert lib.last_kernel() == lib.K_UPDAT_STREAM
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert L.bsmm_debug_u2_trace_copy(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(1024, 8)
    live = t[:, 3] > 0
    n = int(live.sum())
    t = t[:n].astype(np.float64)
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    print("d%d N %d run %d: %d workgroups, ranges walked: %s, chunks of the first range: %s" % (d, N, rep, n, np.unique(t[:, 4]).astype(int).tolist(), np.unique(t[:, 5]).astype(int).tolist()))
    print("  %-28s %8s %8s %8s %8s" % ("us from the first start", "min", "median", "p90", "max"))
    for k, name in ((0, "start"), (1, "first range: loop done"), (2, "first range: sums stored"), (3, "end")):
        v = us(t[:, k])
        print("  %-28s %8.1f %8.1f %8.1f %8.1f" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    print("  per XCD (blockIdx %% 8): first-range loop done  median / max | end  median / max | workgroups with a second range")
    for xcd in range(8):
        m = (np.arange(n) % 8) == xcd
        a, e = us(t[m, 1]), us(t[m, 3])
        print("    xcd %d   %6.1f / %6.1f   |  %6.1f / %6.1f  |  %d" % (xcd, np.median(a), a.max(), np.median(e), e.max(), int((t[m, 4] > 1).sum())))
    # first-range loop time against the item's load (the XCD schedule: workgroup b -> XCD b % 8, uj = b // 8, set = xcd // nparts, item = set_first + uj)
    if rep == 2:
        hw = np.asarray(b._tables_on(torch.device("cuda")).updat_plan.host)
        nsets = int(hw[8]); nparts = 8 // nsets; off = int(hw[6])
        rows = []
        for wg in range(n):
            xcd, uj = wg % 8, wg // 8
            st = xcd // nparts
            item = int(hw[9 + 2 * st]) + uj
            wd = hw[off + item * 84 + 4: off + item * 84 + 84].reshape(16, 5)[:, 0]
            slots = (wd & 15) + ((wd >> 4) & 15)
            simd = max(int(slots[s::4].sum()) for s in range(4))
            rows.append((int(hw[off + item * 84 + 2]), simd, int(slots[:8].sum()), int(slots[8:].sum()), (t[wg, 1] - t[wg, 0]) / 100.0))
        rows = np.array(rows)
        print("  first-range loop time by blocks of the item (mean us over its 4 workgroups | busiest SIMD's blocks | set A / set B blocks):")
        for nbk in sorted(set(rows[:, 0].astype(int))):
            m = rows[:, 0] == nbk
            print("    %3d blocks  %6.1f us  (n=%d)  simd max %4.1f  A %4.1f B %4.1f" % (nbk, rows[m, 4].mean(), m.sum(), rows[m, 1].mean(), rows[m, 2].mean(), rows[m, 3].mean()))
        A = np.stack([np.ones(len(rows)), rows[:, 0], rows[:, 1]], axis=1)
        coef, *_ = np.linalg.lstsq(A, rows[:, 4], rcond=None)
        print("  least squares: us = %.2f + %.3f * blocks + %.3f * busiest-SIMD blocks; corr(time, blocks) = %.3f" % (coef[0], coef[1], coef[2], np.corrcoef(rows[:, 0], rows[:, 4])[0, 1]))
