#!/usr/bin/env python3
"""threads x depth shapes of the decode pool on the C4 batch (256 blobs), long runs: usage: python tools/shape_probe.py "4 4" "8 2" ..."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import corto_amd as ca
blobs, _ = bench.load_blobs(0)
if os.environ.get("KIND") == "irregular":
    from corto_amd import synth
    blobs = [ca.encode(synth.bumpy_sphere_flipped(64, 32, seed=i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(256)]
if os.environ.get("KIND") == "delaunay":                   # bench.py's `realistic` blobs
    from corto_amd import synth
    blobs = [ca.encode(synth.delaunay_disc(2310, seed=i, holes=6 + i % 5), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(256)]
arena = ca.upload_arena(blobs, 0)
arenas = None if os.environ.get("FROM_HOST") in ("1", "2") else [[arena]]      # FROM_HOST=1: every step uploads its blobs from host memory; 2: from ONE pinned buffer, in place
if os.environ.get("FROM_HOST") == "2":
    pin, blobs = ca.pinned_host_arena(blobs)
for sh in sys.argv[1:] or ["4 4", "8 2"]:
    th, dp = (int(x) for x in sh.split())
    pool = ca.Pool([0], threads=th, depth=dp)
    if os.environ.get("FROM_HOST") == "2":
        pool.set_packed_host_blobs(True)
    pool.run([blobs], steps=pool.lanes * 4, warmup=0, arenas=arenas)
    rep, st = pool.run([blobs], steps=1500, warmup=48, arenas=arenas)
    print("%dx%d: %.4f ms/step %.0f Mtri/s host %.0f us/step/thread (plan %.0f) wait %.0f finish %.0f  wall/thread-step %.0f %s" % (th, dp, rep.elapsed_s / 1500 * 1e3, rep.triangles / rep.elapsed_s / 1e6, rep.host_us_per_step, rep.host_plan_us, rep.host_wait_us, rep.host_finish_us, rep.elapsed_s / 1500 * 1e6 * th, pool.warning[:40]), flush=True)
    pool.close()
