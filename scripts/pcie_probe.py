"""Development: pinned host -> device copy bandwidth on this box (the ceiling of the host-resident path)."""
import time, torch
for mb in (64, 256, 1024):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"H2D pinned {mb} MiB: {(mb << 20) / dt / 1e9:.1f} GB/s")
    t0 = time.perf_counter()
    for _ in range(5): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"D2H pinned {mb} MiB: {(mb << 20) / dt / 1e9:.1f} GB/s")
