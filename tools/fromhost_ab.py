#!/usr/bin/env python3
"""SURVEY 8d's primary region (pinned host -> HBM, packed or scattered blobs) against resident inputs:
ms per step (mean / median window), the largest gaps between consecutive completions and the host's share per step.
  python tools/fromhost_ab.py [steps] [threads] [depth]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
import corto_amd as ca
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
blobs, _ = bench.load_blobs(0)
pin, views = ca.pinned_host_arena(blobs)
arena = [[ca.upload_arena(blobs, 0)]]
modes = {"resident": ("0", blobs, arena, False), "pinned": ("0", views, None, True), "scattered": ("0", blobs, None, False)}
want = sys.argv[4].split(",") if len(sys.argv) > 4 else ["resident", "pinned", "scattered", "resident"]
print("GPU_MAX_HW_QUEUES=%s threads=%d depth=%d" % (os.environ.get("GPU_MAX_HW_QUEUES"), threads, depth), flush=True)
for name in want:
    _, items, arenas, packed = modes[name]
    pool = ca.Pool([0], threads=threads, depth=depth)
    pool.set_packed_host_blobs(packed)
    pool.run([items], steps=pool.lanes * 8, warmup=0, arenas=arenas)
    for rep_ in range(2):
        rep, st = pool.run([items], steps=steps, warmup=48, arenas=arenas)
        t = np.asarray(st, dtype=np.float64)
        d = np.diff(t)
        order = np.argsort(d)[::-1][:4]
        w = bench.window_stats(st, pool.lanes)
        print("%-22s %.4f ms/step mean, %.4f median window, %.4f best | host %.0f us/step/thread (plan %.0f wait %.0f harvest %.0f; longest plan %.0f launch %.0f) | largest gaps (us@step): %s" % (
            name, rep.elapsed_s / steps * 1e3, w["median_ms_per_step"], w["best_ms_per_step"], rep.host_us_per_step, rep.host_plan_us, rep.host_wait_us, rep.host_finish_us, rep.host_plan_max_us, rep.host_launch_max_us,
            ", ".join("%.0f@%d" % (d[i] * 1e6, i) for i in order)), flush=True)
    pool.close()
