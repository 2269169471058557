"""Per-kernel HIP-event times of the single-object configs (C2: 128K-vertex mesh, C3: 167K-point cloud), one decode each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
ctx = ca.Context(0); ctx.set_profiling(True)
for key, mesh, kw in (("C2", synth.bumpy_sphere(512, 250, seed=1), dict(normal_prediction=ca.BORDER)), ("C3", synth.point_cloud(578, 289, seed=2), dict(normal_prediction=ca.DIFF))):
    blob = ca.encode(mesh, position_bits=14, uv_bits=12, normal_bits=10, **kw)
    b = ca.Batch(ctx, [blob]); b.allocate_outputs()
    acc = {}
    for i in range(6):
        b.decode(); b.sync()
        if i >= 2:
            for k, v in b.kernel_times().items():
                a = acc.setdefault(k, [0.0, 0]); a[0] += v["ms"]; a[1] += v.get("launches", 1)
    print(key, {k: (round(v[0] / 4, 4), v[1] // 4) for k, v in acc.items()}, "sum", round(sum(v[0] for v in acc.values()) / 4, 3))
    b.close()
