import torch, time
n = 832_000_000
d = torch.empty(n, dtype=torch.uint8, device='cuda')
h = torch.empty(n, dtype=torch.uint8).pin_memory()
for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(name, "%.1f ms  %.1f GB/s" % (t * 1e3, n / t / 1e9))
# both directions at once on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty(n, dtype=torch.uint8, device='cuda'); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
torch.cuda.synchronize(); t = time.perf_counter() - t0
print("duplex %.1f ms  %.1f GB/s each way" % (t * 1e3, n / t / 1e9))
