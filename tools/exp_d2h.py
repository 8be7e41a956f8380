"""Development: device-to-host copy rates (pageable vs pinned) for a 410 MB result."""
import time, torch
n = 9322163
d = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(5)] + [torch.zeros(n, dtype=torch.int32, device="cuda")]
torch.cuda.synchronize()
t0 = time.perf_counter(); pin = [torch.empty(x.shape, dtype=x.dtype, pin_memory=True) for x in d]; t1 = time.perf_counter()
print("pinned alloc %.1f ms" % ((t1 - t0) * 1e3))
for rep in range(3):
    torch.cuda.synchronize(); a = time.perf_counter()
    for x, p in zip(d, pin): p.copy_(x, non_blocking=True)
    torch.cuda.synchronize(); b = time.perf_counter()
    print("D2H pinned: %.2f ms = %.1f GB/s" % ((b - a) * 1e3, n * 44 / (b - a) / 1e9))
pg = [torch.empty(x.shape, dtype=x.dtype) for x in d]
for rep in range(2):
    torch.cuda.synchronize(); a = time.perf_counter()
    for x, p in zip(d, pg): p.copy_(x)
    torch.cuda.synchronize(); b = time.perf_counter()
    print("D2H pageable: %.2f ms = %.1f GB/s" % ((b - a) * 1e3, n * 44 / (b - a) / 1e9))
