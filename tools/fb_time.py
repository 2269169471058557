"""Probe: decode time of meshes whose front outgrows the LDS ring (redone on the HBM front) next to same-size spheres."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import corto_amd as ca
from corto_amd import synth
ctx = ca.Context(0); ctx.set_profiling(True)
cases = [("torus100x50", synth.torus(100, 50, seed=1)), ("sphere100x50", synth.bumpy_sphere(100, 50, seed=1)),
         ("torus200x100", synth.torus(200, 100, seed=2)), ("sphere200x100", synth.bumpy_sphere(200, 100, seed=2)),
         ("strip4000", synth.strip(4000, seed=3)), ("disc40", synth.holey_disc(40, seed=2, color_components=4))]
for name, m in cases:
    blob = ca.encode(m, normal_prediction=ca.BORDER)
    b = ca.Batch(ctx, [blob] * 64); b.allocate_outputs(); b.decode(); b.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        b.decode(); b.sync()
    dt = (time.perf_counter() - t0) / 3
    kt = b.kernel_times()
    print(name, m.nvert, m.nface, "x64: %.2f ms" % (dt * 1e3), "fallbacks", b.stats().topology_fallbacks,
          {k: round(v["ms"], 3) for k, v in kt.items() if "topo" in k}, flush=True)
