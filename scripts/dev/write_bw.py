#!/usr/bin/env python3
"""Dev (round 4): what does a pure WRITE of the cost volume's 251 MB cost on this part?  (The warp kernel's algorithmic traffic is 97 %
stores.)  torch fill / copy kernels as the yardstick."""
import torch


def time_us(run, steps=50, warm=10):
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


x = torch.empty((1, 192, 128, 160, 32), dtype=torch.float16, device="cuda")
y = torch.randn((1, 192, 128, 160, 32), device="cuda").half()
mb = x.numel() * 2 / 1e6
for name, fn, bytes_ in (("fill (write only)", lambda: x.fill_(1.0), mb), ("zero_ (write only)", lambda: x.zero_(), mb),
                         ("copy_ (read + write)", lambda: x.copy_(y), 2 * mb), ("mul_ in place (read + write)", lambda: y.mul_(1.0001), 2 * mb)):
    ts = sorted(time_us(fn) for _ in range(5))
    print(f"{name}: {ts[2]:.1f} us for {bytes_:.0f} MB = {bytes_ / ts[2] * 1e-3 * 1e3:.0f} GB/s", flush=True)
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
ts = sorted(time_us(lambda: big.zero_(), 20, 5) for _ in range(3))
print(f"zero_ of 1 GiB: {ts[1]:.1f} us = {1073.74 / ts[1] * 1e3:.0f} GB/s")
