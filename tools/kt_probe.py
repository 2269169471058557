"""Probe: per-kernel HIP-event times of one C4 batch decoded unpipelined, no verification (the A/B instrument of kernel changes: two builds, tools/ab_build.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import corto_amd as ca
import bench
blobs, _z = bench.load_blobs(0)
ctx = ca.Context(0); ctx.set_profiling(True)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
acc = {}
N = 12
for i in range(N + 3):
    b.decode(); b.sync(raise_on_error=False)
    if i >= 3:
        for k, v in b.kernel_times().items():
            a = acc.setdefault(k, [0.0, 0]); a[0] += v["ms"]; a[1] += v.get("launches", 1)
print(os.environ.get("CORTO_HIP_LIB_PATH", "-"), {k: (round(v[0] / N, 4), v[1] // N) for k, v in acc.items()})
