import torch, time
dev = torch.device("cuda:0")
for mb in (8, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(10):
            d.copy_(h, non_blocking=True)
    s.synchronize()
    dt = time.perf_counter() - t0
    print(f"H2D pinned {mb} MB x10: {10 * mb / 1024 / dt:.1f} GB/s")
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(10):
            h.copy_(d, non_blocking=True)
    s.synchronize()
    print(f"D2H pinned {mb} MB x10: {10 * mb / 1024 / (time.perf_counter() - t0):.1f} GB/s")
