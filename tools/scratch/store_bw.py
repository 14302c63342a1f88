"""what a pure store stream / copy achieves on this box (torch fill_ / copy_), for sizing store-bound kernels"""
import statistics, torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)
for mb in (26, 52, 104, 208, 416):
    x = torch.empty(mb * 500000, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x))
    print("%4d MB  fill %6.1f us = %5.2f TB/s   copy %6.1f us = %5.2f TB/s (read + write)" % (mb, tf, mb / tf, tc, 2 * mb / tc))
