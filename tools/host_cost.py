"""Probe: host wall time of each C-ABI call of one pipelined step (single thread, depth 4)."""
import os, sys, time, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import corto_amd as ca
import bench
blobs, _z = bench.load_blobs(0)
L = ca.lib(); n = len(blobs)
ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in blobs]); lens = np.array([len(x) for x in blobs], dtype=np.uint32)
arena = ca.upload_arena(blobs, 0)
depth = 4
ctxs = [ca.Context(0) for _ in range(depth)]
keeps = []
for c in ctxs:
    b = ca.Batch(c, blobs, device_arena=arena); b.allocate_outputs(); keeps.append((b, b._keep))
status = np.zeros(n, dtype=np.int32)
acc = collections.defaultdict(float)
def timed(name, f, *a):
    t0 = time.perf_counter(); r = f(*a); acc[name] += time.perf_counter() - t0; return r
pend = [None] * depth
steps = 80
for i in range(steps + 8):
    if i == 8: acc.clear(); t_all = time.perf_counter()
    k = i % depth
    if pend[k] is not None:
        timed("sync", L.crthip_batch_sync, pend[k], status.ctypes.data_as(C.c_void_p)); timed("destroy", L.crthip_batch_destroy, pend[k])
    h = C.c_void_p()
    buf, binds, index_ptrs, index_fmt = keeps[k][1]
    timed("create", L.crthip_batch_create, ctxs[k].handle, n, ptrs, lens.ctypes.data_as(C.c_void_p), C.c_void_p(arena.data_ptr()), C.byref(h))
    timed("bind_all", L.crthip_batch_bind_all, h, binds, index_ptrs, index_fmt.ctypes.data_as(C.c_void_p))
    timed("decode", L.crthip_batch_decode, h)
    pend[k] = h
tot = time.perf_counter() - t_all
print("ms/step %.3f" % (tot / steps * 1e3), {k: round(v / steps * 1e6, 1) for k, v in acc.items()}, "us per step")
