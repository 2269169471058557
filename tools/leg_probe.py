#!/usr/bin/env python3
"""The side legs of bench.py (one dictionary per stream; irregular connectivity) at several run lengths, on a pool of their own:
usage (GPU box): python tools/leg_probe.py [steps ...]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import corto_amd as ca
from corto_amd import synth

steps = [int(x) for x in sys.argv[1:]] or [120, 1200]
blobs, _ = bench.load_blobs(0)
iblobs = [ca.encode(synth.bumpy_sphere_flipped(64, 32, seed=i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(256)]
for name, items, share in (("regular shared", blobs, None), ("regular, a dictionary per stream, one kernel", blobs, "0"), ("regular, a dictionary per stream, two kernels", blobs, "2"), ("irregular shared", iblobs, None), ("irregular, a dictionary per stream, two kernels", iblobs, "2")):
    if share is not None:
        os.environ["CORTO_TUN_SHARE"] = share
    pool = ca.Pool([0], threads=4, depth=4)
    os.environ.pop("CORTO_TUN_SHARE", None)
    arena = [[ca.upload_arena(items, 0)]]
    pool.run([items], steps=4 * pool.lanes, warmup=0, arenas=arena)
    for n in steps:
        rep, st = pool.run([items], steps=n, warmup=2 * pool.lanes, arenas=arena)
        print("%-46s %5d steps: %8.1f Mtri/s  %.4f ms/step  host %.0f us/step/thread  fallbacks %d" % (name, n, rep.triangles / rep.elapsed_s / 1e6, rep.elapsed_s / n * 1e3, rep.host_us_per_step, rep.topology_fallbacks), flush=True)
    pool.close()
