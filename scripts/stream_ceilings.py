#!/usr/bin/env python3
"""Achievable HBM rates on this device for the traffic patterns of K1 (3 read streams) and K2
(2 read + 1 write stream), measured with plain torch elementwise kernels on planes of the
bench's size: the practical ceilings to read roofline fractions against."""
import time, torch
n = 20_000_000 * 160
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device='cuda')
b = torch.randint(0, 255, (n,), dtype=torch.uint8, device='cuda')
c = torch.empty_like(a)
a32, b32, c32 = a.view(torch.int32), b.view(torch.int32), c.view(torch.int32)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, f, nbytes in (
        ('copy (1 read + 1 write)', lambda: c32.copy_(a32), 2 * n),
        ('add int32 (2 read + 1 write)', lambda: torch.add(a32, b32, out=c32), 3 * n),
        ('add uint8 (2 read + 1 write)', lambda: torch.add(a, b, out=c), 3 * n),
        ('sum int32 (1 read)', lambda: a32.sum(), n),
        ('eq+sum (2 read)', lambda: (a32 == b32).sum(), 2 * n),
):
    dt = timeit(f)
    print('%-32s %.3f ms  %.0f GB/s' % (name, dt * 1e3, nbytes / dt / 1e9), flush=True)
