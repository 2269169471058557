#!/usr/bin/env python3
"""K-DELTA: scan passes (default) against the flag-driven walk alone ($CORTO_EXP_DELTA_WALK=1) on batches of 256 meshes of one kind."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import corto_amd as ca
from corto_amd import synth
kinds = {"grid 64x32 (C4 unit)": lambda i: synth.bumpy_sphere(64, 32, seed=i), "torus 48x24": lambda i: synth.torus(48, 24, seed=i),
         "closed sphere 40x24": lambda i: synth.closed_sphere(40, 24, seed=i), "holey disc 40": lambda i: synth.holey_disc(40, seed=i),
         "shuffled grid 32x16": lambda i: synth.shuffled(synth.bumpy_sphere(32, 16, seed=i), seed=i)}
if os.environ.get("KIND"):
    kinds = {k: v for k, v in kinds.items() if os.environ["KIND"] in k}
ctx = ca.Context(0)
ctx.set_profiling(True)
for name, make in kinds.items():
    blobs = [ca.encode(make(i), normal_prediction=ca.BORDER) for i in range(64)] * 4
    b = ca.Batch(ctx, blobs); b.allocate_outputs()
    best = {}
    for _ in range(5):
        b.decode(); assert (b.sync() == 0).all()
        for k, v in b.kernel_times().items():
            best[k] = min(best.get(k, 1e9), v["ms"])
    st = b.stats()
    print("   clers symbols/blob %d, topology fallbacks %d of %d, lds scale %d" % (st.clers_symbols // len(blobs), st.topology_fallbacks, len(blobs), st.topology_scale))
    print("%-24s delta %.4f ms  topology %.4f  normals %.4f  (walk only: %s)" % (name, best.get("delta_mesh", 0), best.get("topology_lds", 0), best.get("normal_blob", 0), os.environ.get("CORTO_EXP_DELTA_WALK", "0")), flush=True)
    b.close()
