#!/usr/bin/env python3
"""Per-kernel HIP-event times of one C4 batch, unpipelined, on a context configured like the pool's (single stream: the LDS-lean
layouts); $SHARE=0: one dictionary per stream; $MESH=flipped for irregular connectivity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
import corto_amd as ca
from corto_amd import synth
if os.environ.get("SHARE") == "0":
    os.environ["CORTO_TUN_SHARE"] = "0"
if os.environ.get("MESH") == "flipped":
    blobs = [ca.encode(synth.bumpy_sphere_flipped(64, 32, seed=i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(256)]
else:
    blobs, _ = bench.load_blobs(0)
for single in (True, False):
    ctx = ca.Context(0); ctx.set_profiling(True); ctx.set_single_stream(single)
    arena = ca.upload_arena(blobs, 0)
    b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
    acc = {}
    N = 8
    for i in range(N + 3):
        b.decode(); b.sync()
        if i >= 3:
            for k, v in b.kernel_times().items():
                a = acc.setdefault(k, [0.0, 0]); a[0] += v["ms"]; a[1] += v.get("launches", 1)
    print("single_stream" if single else "two streams  ", {k: round(v[0] / N * 1e3, 1) for k, v in acc.items()}, "us; sum", round(sum(v[0] for v in acc.values()) / N * 1e3, 1), flush=True)
    b.close(); ctx.close()
