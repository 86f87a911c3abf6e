"""Development: pinned H2D / D2H copy rates by transfer size on this box (the ceiling of the host-resident path)."""
import time
import torch
for mib in (8, 32, 64, 256, 1024):
    n = mib << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for direction in ("h2d", "d2h"):
        torch.cuda.synchronize()
        reps = max(2, 2048 // mib)
        t0 = time.perf_counter()
        for _ in range(reps):
            if direction == "h2d":
                d.copy_(h, non_blocking=True)
            else:
                h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{direction} {mib:5d} MiB: {n / dt / 1e9:6.1f} GB/s", flush=True)
