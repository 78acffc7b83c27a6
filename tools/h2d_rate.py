import torch, time
dev = torch.device("cuda", 0)
for mb in (0.064, 1, 4, 16, 64, 128, 256):
    n = int(mb * (1 << 20))
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        reps = max(4, int(2e9 / n)) if n < (64 << 20) else 12
        reps = min(reps, 2000)
        t0 = time.perf_counter()
        for _ in range(reps):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t0
    print("%8.3f MB x %4d: %.1f GB/s  (%.1f us per copy)" % (mb, reps, n * reps / dt / 1e9, dt / reps * 1e6))
