"""How does hipMemcpyAsync D2H / H2D of pinned memory execute on this box (SDMA engine or a blit kernel), how fast is
it, and what does it do to an HBM-bound kernel running beside it?"""
import os, sys, time, threading
import torch
dev = torch.device("cuda", 0)
MB = int(os.environ.get("PROBE_MB", "16"))
n = MB * (1 << 20) // 4
d = torch.randn(n, device=dev)
h = torch.empty(n, dtype=torch.float32).pin_memory()
big = torch.randn(256 << 20, device=dev)            # 1 GiB
out = torch.empty_like(big)
side = torch.cuda.Stream()
torch.cuda.synchronize()

def copies(k, direction):
    with torch.cuda.stream(side):
        for _ in range(k):
            if direction == "d2h":
                h.copy_(d, non_blocking=True)
            else:
                d.copy_(h, non_blocking=True)

def hbm(k):
    for _ in range(k):
        torch.add(big, 1.0, out=out)

for direction in ("d2h", "h2d"):
    copies(3, direction); torch.cuda.synchronize()
    t = time.perf_counter(); copies(20, direction); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{direction} alone: {20 * MB / 1024 / dt:.1f} GiB/s ({1e3 * dt / 20:.3f} ms per {MB} MiB)")
hbm(3); torch.cuda.synchronize()
t = time.perf_counter(); hbm(40); torch.cuda.synchronize(); base = time.perf_counter() - t
print(f"add 1 GiB alone: {1e3 * base / 40:.3f} ms ({2 * 40 / base:.0f} GiB/s)")
for direction in ("d2h", "h2d"):
    torch.cuda.synchronize()
    t = time.perf_counter()
    copies(60, direction)
    hbm(40)
    torch.cuda.current_stream().synchronize()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"add 1 GiB beside {direction} copies: {1e3 * dt / 40:.3f} ms  (x{dt / base:.2f})")
