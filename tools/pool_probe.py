#!/usr/bin/env python3
"""How evenly a crthip_pool's step completions are spaced for a (threads x depth) shape, and what that does to a short timed
region: usage (GPU box): python tools/pool_probe.py [threads depth]..."""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import corto_amd as ca

blobs, _ = bench.load_blobs(0)
arena = ca.upload_arena(blobs, 0)
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(2, 3), (6, 1), (3, 2), (4, 1), (8, 1)]
for th, dp in shapes:
    pool = ca.Pool([0], threads=th, depth=dp)
    pool.run([blobs], steps=pool.lanes * 2, warmup=0, arenas=[[arena]])
    if os.environ.get("PROBE_LONG"):                       # one long run and nothing else (tools/prof_pipe.sh: the trace's steady state)
        n = int(os.environ["PROBE_LONG"])
        rep, st = pool.run([blobs], steps=n, warmup=48, arenas=[[arena]])
        print("threads %d depth %d: %d steps %.4f ms/step (%.1f Mtri/s)" % (th, dp, n, rep.elapsed_s / n * 1e3, rep.triangles / rep.elapsed_s / 1e6), flush=True)
        pool.close()
        continue
    rep, st = pool.run([blobs], steps=480, warmup=48, arenas=[[arena]])
    long_ms = rep.elapsed_s / 480 * 1e3
    d = np.diff(np.concatenate([[0], st])) * 1e3
    short = []
    for trial in range(12):
        r2, s2 = pool.run([blobs], steps=20, warmup=3 + trial % 4, arenas=[[arena]])
        short.append(r2.elapsed_s / 20 * 1e3)
    print("threads %d depth %d: 480 steps %.4f ms/step | gaps ms: p10 %.3f p50 %.3f p90 %.3f max %.3f | 20-step runs: min %.3f med %.3f max %.3f" % (
        th, dp, long_ms, np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max(), min(short), np.median(short), max(short)), flush=True)
    pool.close()
