#!/usr/bin/env python3
"""from-host pipelined steps: where the time goes when the mean step is far above the median one - the largest gaps between consecutive
completions of a long run (FROM_HOST=1: scattered pageable blobs, gathered by the library; 2: one pinned buffer, uploaded in place)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import corto_amd as ca
blobs, _ = bench.load_blobs(0)
mode = os.environ.get("FROM_HOST", "1")
if mode == "2":
    pin, blobs = ca.pinned_host_arena(blobs)
pool = ca.Pool([0], threads=4, depth=4)
if mode == "2":
    pool.set_packed_host_blobs(True)
pool.run([blobs], steps=pool.lanes * 4, warmup=0, arenas=None)
for rep_ in range(3):
    rep, st = pool.run([blobs], steps=3000, warmup=48, arenas=None)
    t = np.asarray(st, dtype=np.float64)
    d = np.diff(t)
    order = np.argsort(d)[::-1][:6]
    print("mode %s: %.4f ms/step mean, median gap %.1f us, largest gaps (us @ step): %s" % (mode, rep.elapsed_s / 3000 * 1e3, np.median(d) * 1e6,
          ", ".join("%.0f@%d" % (d[i] * 1e6, i) for i in order)), flush=True)
pool.close()
