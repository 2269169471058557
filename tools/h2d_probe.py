#!/usr/bin/env python3
"""What the box's PCIe gives a pinned-host -> HBM copy of a C4 batch's .crt bytes (3.7 MB) - the ceiling of SURVEY 8d's primary region:
one stream back to back, then S streams at once; also 16 MB and 64 MB chunks.  ($HSA_ENABLE_SDMA=0 in the environment: blit kernels.)"""
import os, sys, time
import torch
dev = torch.device("cuda", 0)
for mb in (3.73, 16.0, 64.0):
    n = int(mb * 1e6)
    for S in (1, 4, 16):
        src = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(S)]
        dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(S)]
        st = [torch.cuda.Stream(dev) for _ in range(S)]
        for k in range(S):
            with torch.cuda.stream(st[k]):
                dst[k].copy_(src[k], non_blocking=True)
        torch.cuda.synchronize()
        reps = max(4, int(400e6 / n / S))
        t0 = time.perf_counter()
        for r in range(reps):
            for k in range(S):
                with torch.cuda.stream(st[k]):
                    dst[k].copy_(src[k], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("H2D %6.2f MB x %2d streams: %6.1f GB/s (%.1f us per copy)" % (mb, S, reps * S * n / dt / 1e9, dt / (reps * S) * 1e6), flush=True)
