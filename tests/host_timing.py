import sys, os, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, corto_amd as ca, torch
z = np.load('/root/repo/tests/golden/c4_blobs16.npz')
blobs = [ca.aligned_blob(z["crt_%02d" % (s % 16)]) for s in range(256)]
ctx = ca.Context(0)
arena = ca.upload_arena(blobs)
b0 = ca.Batch(ctx, blobs, device_arena=arena); b0.allocate_outputs()
buf, binds, index_ptrs, index_fmt = b0._keep
L = ca.lib(); n = len(blobs)
ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in blobs]); lens = np.array([len(x) for x in blobs], dtype=np.uint32)
status = np.zeros(n, dtype=np.int32)
T = np.zeros(5)
for it in range(23):
    t0 = time.perf_counter(); h = C.c_void_p()
    L.crthip_batch_create(ctx.handle, n, ptrs, lens.ctypes.data_as(C.c_void_p), C.c_void_p(arena.data_ptr()), C.byref(h)); t1 = time.perf_counter()
    L.crthip_batch_bind_all(h, binds, index_ptrs, index_fmt.ctypes.data_as(C.c_void_p)); t2 = time.perf_counter()
    L.crthip_batch_decode(h); t3 = time.perf_counter()
    L.crthip_batch_sync(h, status.ctypes.data_as(C.c_void_p)); t4 = time.perf_counter()
    L.crthip_batch_destroy(h); t5 = time.perf_counter()
    if it >= 3: T += [t1-t0, t2-t1, t3-t2, t4-t3, t5-t4]
print("ms: create %.3f bind %.3f decode(launch) %.3f sync %.3f destroy %.3f" % tuple(T/20*1e3))
