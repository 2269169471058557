#!/usr/bin/env python3
"""Is the pipelined step bound by per-step overheads (launches, host) or by GPU work?  Same pool, batches of 64 .. 1024 blobs."""
import os, sys
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import corto_amd as ca
b0, _ = bench.load_blobs(0); b1, _ = bench.load_blobs(256); b2, _ = bench.load_blobs(512); b3, _ = bench.load_blobs(768)
allb = b0 + b1 + b2 + b3
th, dp = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 3)
for n in (64, 128, 256, 512, 1024):
    blobs = allb[:n]
    arena = ca.upload_arena(blobs, 0)
    pool = ca.Pool([0], threads=th, depth=dp)
    pool.run([blobs], steps=pool.lanes * 4, warmup=0, arenas=[[arena]])
    rep, st = pool.run([blobs], steps=1000, warmup=48, arenas=[[arena]])
    ms = rep.elapsed_s / 1000 * 1e3
    print("%4d blobs/batch (%dx%d): %.4f ms/step  %.2f us/blob  %.0f Mtri/s" % (n, th, dp, ms, ms * 1e3 / n, n * 4096 / ms / 1e3), flush=True)
    pool.close()
