"""HBM write / copy rates seen by simple torch kernels on this box (context for the store-bound attention kernels)."""
import torch
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (192, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    t = timeit(lambda: a.zero_()); print("zero_  %5d MB: %.3f ms  %.2f TB/s (write)" % (mb, t, mb * 1.048576e6 / t / 1e9))
    t = timeit(lambda: b.copy_(a)); print("copy_  %5d MB: %.3f ms  %.2f TB/s (read+write)" % (mb, t, 2 * mb * 1.048576e6 / t / 1e9))
    t = timeit(lambda: a.sum()); print("sum    %5d MB: %.3f ms  %.2f TB/s (read)" % (mb, t, mb * 1.048576e6 / t / 1e9))
