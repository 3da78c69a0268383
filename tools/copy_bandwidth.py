#!/usr/bin/env python3
"""Device copy bandwidth of the box (report the HBM fraction against the nominal 8 TB/s AND against what a
copy kernel reaches here).  Read + write bytes per second of a 2 GiB device-to-device copy and of a 3-array triad."""
import time, torch
n = 1 << 29                                    # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a); c = torch.empty_like(a)
def bw(fn, nbytes, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12
print(f"copy   (read + write): {bw(lambda: b.copy_(a), 2 * 4 * n):.2f} TB/s")
print(f"triad  (2 reads + write): {bw(lambda: torch.add(a, b, alpha=2.0, out=c), 3 * 4 * n):.2f} TB/s")
print(f"fill   (write): {bw(lambda: c.fill_(1.0), 4 * n):.2f} TB/s")
