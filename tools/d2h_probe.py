#!/usr/bin/env python3
"""What PCIe gives the secondary region's copies: 32 MB HBM -> pinned host, back to back on ONE stream, round-robin over 20 streams, and over 20 streams with a
3.7 MB host -> HBM copy in front of each (the primary region's upload going the other way)."""
import torch, time
N = 32047104
dev = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(20)]
host = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(20)]
up_h = [torch.empty(3736032, dtype=torch.uint8).pin_memory() for _ in range(20)]
up_d = [torch.empty(3736032, dtype=torch.uint8, device="cuda") for _ in range(20)]
streams = [torch.cuda.Stream() for _ in range(20)]
def run(nstreams, with_upload, reps=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(reps):
        k = r % nstreams
        with torch.cuda.stream(streams[k]):
            if with_upload: up_d[k].copy_(up_h[k], non_blocking=True)
            host[k].copy_(dev[k], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return N * reps / dt / 1e9, dt / reps * 1e3
for ns, up in ((1, False), (20, False), (20, True), (4, True)):
    run(ns, up, 10)
    g, ms = run(ns, up)
    print("%2d streams%s: %.1f GB/s, %.3f ms a copy" % (ns, " + 3.7 MB upload each" if up else "", g, ms))
