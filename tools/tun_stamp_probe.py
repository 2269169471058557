"""Where a short Tunstall stream's time goes (k_tun_stream): needs the library built with CORTO_BUILD_DEFINES=CORTO_TUN_STAMPS
(python -m corto_amd.build --force); phases: seed, dictionary growth, survivors spelled out, decode."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import corto_amd as ca, bench
blobs, _z = bench.load_blobs(0)
ctx = ca.Context(0)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
for i in range(4): b.decode(); b.sync()
out = np.zeros(8*4096, dtype=np.uint64)
L = ca.lib(); L.crthip_debug_tun_stamps.argtypes = [C.c_void_p]
print("rc", L.crthip_debug_tun_stamps(out.ctypes.data_as(C.c_void_p)))
o = out.reshape(4096, 8).astype(np.int64)
o = o[o[:, 0] > 0]
t0 = o[:, 0].min()
ph = np.diff(o[:, :5], axis=1) / 100.0   # us at 100 MHz
print("streams", len(o), "start spread us", (o[:, 0].max() - t0) / 100.0, "end max us", (o[:, 4].max() - t0) / 100.0)
print("phase means us [seed, grow, final, decode]:", ph.mean(0).round(2), "max:", ph.max(0).round(2))
tot = (o[:, 4] - o[:, 0]) / 100.0
worst = np.argsort(-tot)[:12]
for w in worst: print("n=%d csize=%d size=%d" % tuple(o[w, 5:8]), "phases", ph[w].round(2), "start", (o[w, 0] - t0) / 100.0)
import collections
byn = collections.defaultdict(list)
for r, p in zip(o, ph): byn[int(r[5])].append(p)
for n in sorted(byn): print("n=%3d count %4d mean" % (n, len(byn[n])), np.mean(byn[n], 0).round(2))
