"""Calibration: what a plain device-to-device streaming copy achieves on this box (torch)."""
import torch, time
for n in (51_000_000, 205_000_000):  # doubles: 0.41 GB and 1.64 GB
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {n*8/1e9:.2f} GB: {ms:.3f} ms -> {2*n*8/ms/1e9:.2f} TB/s (read+write)")
    e0.record()
    for _ in range(10): y.zero_()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill {n*8/1e9:.2f} GB: {ms:.3f} ms -> {n*8/ms/1e9:.2f} TB/s (write only)")
