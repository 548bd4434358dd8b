#!/usr/bin/env python3
"""Measured HBM ceilings on this GPU (torch elementwise kernels, not part of the product): pure write (fill),
copy (read+write), pure read (sum).  Used to judge how far each gain kernel is from what the memory system gives."""
import time
import torch

n = 6 * 2 ** 30 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


gb = n * 8 / 1e9
print("fill  (write)      %.0f GB/s" % (gb / timeit(lambda: x.fill_(1.5))))
print("copy  (read+write) %.0f GB/s" % (2 * gb / timeit(lambda: y.copy_(x))))
print("sum   (read)       %.0f GB/s" % (gb / timeit(lambda: x.sum())))
print("add   (2r+1w)      %.0f GB/s" % (3 * gb / timeit(lambda: torch.add(x, y, out=y))))
