#!/usr/bin/env python3
"""A long run of the primary region (pinned host -> HBM, 5 x 4 contexts): ms a step per ten-second slice, device memory and host RSS before / after, failed blobs,
and the outputs of every context's last step against the oracle.   python tools/soak.py [seconds]"""
import os, sys, time, resource
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
import corto_amd as ca
from oracle import oracle as oc
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
blobs, _ = bench.load_blobs(0)
pin, views = ca.pinned_host_arena(blobs)
pool = ca.Pool([0], threads=5, depth=4)
pool.set_packed_host_blobs(True)
pool.run([views], steps=pool.lanes * 8, warmup=0)
free0 = torch.cuda.mem_get_info(0)[0]; rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
t_end = time.time() + seconds
total = failed = 0
while time.time() < t_end:
    rep, st = pool.run([views], steps=100000, warmup=0)
    total += rep.steps; failed += rep.failed_blobs
    print("%8d steps  %.4f ms/step  failed blobs %d  fallbacks %d" % (total, rep.elapsed_s / rep.steps * 1e3, rep.failed_blobs, rep.topology_fallbacks), flush=True)
free1 = torch.cuda.mem_get_info(0)[0]; rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("device memory free %.1f -> %.1f MiB, host max RSS %.1f -> %.1f MiB, failed blobs %d of %d" % (free0 / 2**20, free1 / 2**20, rss0 / 1024, rss1 / 1024, failed, total * len(blobs)))
bad = 0
for lane in range(pool.lanes):
    for i in (0, 17, 255):
        ref = oc.decode(ca.aligned_blob(blobs[i]), color_components=4)
        for k in ("position", "normal", "color", "uv", "index"):
            got = pool.lane_read(lane, i, k, ref[k].dtype, ref[k].size)
            bad += got.tobytes() != ref[k].tobytes()
print("outputs of the last steps against the oracle: %d mismatching arrays over %d contexts" % (bad, pool.lanes))
pool.close()
