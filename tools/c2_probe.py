"""Config C2 (one 128K-vertex mesh) decoded alone, un-profiled: wall ms a decode - on a lone (two-stream) context and on a single-stream one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import corto_amd as ca
from corto_amd import synth
blob = ca.encode(synth.bumpy_sphere(512, 250, seed=1), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
for single in (False, True):
    ctx = ca.Context(0)
    if single: ctx.set_single_stream(True)
    b = ca.Batch(ctx, [blob]); b.allocate_outputs()
    for i in range(3): b.decode(); b.sync()
    t = time.perf_counter()
    for i in range(10): b.decode(); b.sync()
    print("C2 wall ms %.3f (%s)" % ((time.perf_counter() - t) / 10 * 1e3, "single stream" if single else "two streams"))
    b.close(); ctx.close()
