"""Probe: `threads` host threads, each pipelining its own `depth` contexts (ctypes releases the GIL inside the C ABI)."""
import os, sys, time, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
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
for spec in os.environ.get("SPECS", "1x4,2x2,2x3,2x4,3x2,4x2").split(","):
    T, depth = [int(x) for x in spec.split("x")]
    groups = []
    for t in range(T):
        ctxs = [ca.Context(0) for _ in range(depth)]
        keeps = []
        for c in ctxs:
            b = ca.Batch(c, blobs, device_arena=arena); b.allocate_outputs(); keeps.append((b, b._keep))
        groups.append((ctxs, keeps))
    def worker(g, steps):
        ctxs, keeps = g
        status = np.zeros(n, dtype=np.int32)
        pend = [None] * depth
        def finish(h):
            ca._check(L.crthip_batch_sync(h, status.ctypes.data_as(C.c_void_p))); L.crthip_batch_destroy(h)
        for i in range(steps):
            k = i % depth
            if pend[k] is not None: finish(pend[k])
            h = C.c_void_p()
            buf, binds, index_ptrs, index_fmt = keeps[k][1]
            ca._check(L.crthip_batch_create(ctxs[k].handle, n, ptrs, lens.ctypes.data_as(C.c_void_p), C.c_void_p(arena.data_ptr()), C.byref(h)))
            ca._check(L.crthip_batch_bind_all(h, binds, index_ptrs, index_fmt.ctypes.data_as(C.c_void_p)))
            ca._check(L.crthip_batch_decode(h))
            pend[k] = h
        for h in pend:
            if h is not None: finish(h)
        assert (status == 0).all()
    def run(steps_total):
        ths = [threading.Thread(target=worker, args=(g, steps_total // T)) for g in groups]
        for t in ths: t.start()
        for t in ths: t.join()
    run(8 * T); torch.cuda.synchronize()
    steps = 48
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("threads x depth", spec, "ms/step %.3f" % (dt / steps * 1e3), "Mtri/s %.1f" % (ntri * steps / dt / 1e6), flush=True)
    del groups
