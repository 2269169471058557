"""Probe: one large mesh (config C2, 128 512 verts / 256 000 tris) and one large cloud (C3, 167 042 points) through the batch API:
per-kernel device times.  These go down the HBM-front topology path / the scan-based cloud path, not the LDS small-blob path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
ctx = ca.Context(0)
ctx.set_profiling(True)
for name, mesh, kw in (("C2 mesh 256K tris", synth.bumpy_sphere(512, 250, seed=1), dict(normal_prediction=ca.BORDER)),
                       ("C3 cloud 167K pts", synth.point_cloud(578, 289, seed=2), dict(normal_prediction=ca.DIFF))):
    blob = ca.encode(mesh, position_bits=14, uv_bits=12, normal_bits=10, **kw)
    b = ca.Batch(ctx, [blob]); b.allocate_outputs()
    b.decode(); b.sync()
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        b.decode(); b.sync()
    dt = (time.perf_counter() - t0) / n
    kt = b.kernel_times()
    tri = mesh.nface or 0
    print(name, "blob %d B  %.2f ms/decode  %.1f Mtri/s  %.1f Mvert/s" % (len(blob), dt * 1e3, tri / dt / 1e6, mesh.nvert / dt / 1e6))
    print("   ", {k: round(v["ms"], 3) for k, v in kt.items()}, "fallbacks", b.stats().topology_fallbacks)
