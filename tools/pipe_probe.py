"""Probe: throughput of the C4 batch with `depth` contexts in flight (steps pipelined over HIP streams)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import corto_amd as ca
import bench
blobs, _z = bench.load_blobs(0)
L = ca.lib()
n = len(blobs)
ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in blobs])
lens = np.array([len(x) for x in blobs], dtype=np.uint32)
arena = ca.upload_arena(blobs, 0)
ntri = 256 * 4096
for depth in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4").split(",")]:
    ctxs = [ca.Context(0) for _ in range(depth)]
    keeps = []
    for c in ctxs:
        b = ca.Batch(c, blobs, device_arena=arena); b.allocate_outputs(only=set(os.environ['ONLY'].split(',')) if os.environ.get('ONLY') else None); keeps.append((b, b._keep))
    status = np.zeros(n, dtype=np.int32)
    def launch(k):
        h = C.c_void_p()
        buf, binds, index_ptrs, index_fmt = keeps[k][1]
        ca._check(L.crthip_batch_create(ctxs[k].handle, n, ptrs, lens.ctypes.data_as(C.c_void_p), C.c_void_p(arena.data_ptr()), C.byref(h)))
        ca._check(L.crthip_batch_bind_all(h, binds, index_ptrs, index_fmt.ctypes.data_as(C.c_void_p)))
        ca._check(L.crthip_batch_decode(h))
        return h
    def finish(h):
        ca._check(L.crthip_batch_sync(h, status.ctypes.data_as(C.c_void_p)))
        L.crthip_batch_destroy(h)
    def run(steps):
        pend = [None] * depth
        for i in range(steps):
            k = i % depth
            if pend[k] is not None: finish(pend[k])
            pend[k] = launch(k)
        for h in pend:
            if h is not None: finish(h)
    run(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps = 40; run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("depth", depth, "ms/step %.3f" % (dt / steps * 1e3), "Mtri/s %.1f" % (ntri * steps / dt / 1e6), flush=True)
    assert (status == 0).all()
