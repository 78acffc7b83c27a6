import torch, time
n = 800 * 1024 * 1024
d = torch.empty(n, dtype=torch.uint8, device="cuda"); d.fill_(3); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); h = torch.empty(n, dtype=torch.uint8); h.copy_(d); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fresh pageable: %.1f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9)); del h
h = torch.empty(n, dtype=torch.uint8); h.zero_()
for rep in range(3):
    t0 = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pre-faulted pageable: %.1f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
t0 = time.perf_counter(); p = torch.empty(n, dtype=torch.uint8, pin_memory=True); print("pin alloc %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter(); p.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pinned: %.1f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
# chunked through a small pinned ring + CPU memcpy into pageable (2 x 32 MB)
ring = [torch.empty(32 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
ev = [torch.cuda.Event() for _ in range(2)]
h2 = torch.empty(n, dtype=torch.uint8); h2.zero_()
for rep in range(2):
    t0 = time.perf_counter()
    k = 0; c = 32 << 20
    ring[0].copy_(d[0:c], non_blocking=True); ev[0].record()
    for off in range(0, n, c):
        nxt = off + c
        if nxt < n:
            ring[(k + 1) & 1].copy_(d[nxt:nxt + c], non_blocking=True); ev[(k + 1) & 1].record()
        ev[k & 1].synchronize()
        h2[off:off + c].copy_(ring[k & 1])
        k += 1
    dt = time.perf_counter() - t0
    print("ring 2x32MB + one-thread memcpy: %.1f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
