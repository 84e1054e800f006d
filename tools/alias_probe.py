#!/usr/bin/env python3
"""Do power-of-two distances between concurrently streamed vectors cost HBM bandwidth?
Copies / triads between 128 MiB vectors carved out of one allocation at distance 128 MiB + skew."""
import torch

n = 1 << 24  # doubles: 128 MiB
buf = torch.zeros(6 * n + (1 << 22), dtype=torch.float64, device="cuda")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for skew_kb in (0, 4, 68, 260, 1028, 4100, 16388):
    sk = skew_kb * 128  # doubles per KiB = 128
    a = buf[0:n]
    b = buf[n + sk:2 * n + sk]
    c = buf[2 * n + 2 * sk:3 * n + 2 * sk]
    d = buf[3 * n + 3 * sk:4 * n + 3 * sk]
    t_copy = timeit(lambda: b.copy_(a))
    t_add = timeit(lambda: torch.add(a, b, out=c))                 # 2 reads + 1 write
    t_4 = timeit(lambda: torch.addcmul(a, b, c, out=d))            # 3 reads + 1 write
    print("skew %6d KiB: copy %.1f GB/s  add(2r+1w) %.1f GB/s  addcmul(3r+1w) %.1f GB/s" % (
        skew_kb, 2 * n * 8 / t_copy / 1e6, 3 * n * 8 / t_add / 1e6, 4 * n * 8 / t_4 / 1e6))
