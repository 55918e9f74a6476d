"""achievable HBM bandwidth on this box: device-to-device copy (1 read + 1 write per byte) and fill (write only)"""
import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for name, fn, bytes_ in (("copy r+w", lambda: b.copy_(a), 2 * n), ("fill w", lambda: a.fill_(1), n), ("sum r", lambda: a.view(torch.int32).sum(), n)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name}: {bytes_ / ms / 1e6:.0f} GB/s ({ms:.3f} ms)", flush=True)
