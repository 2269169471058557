"""How many CLERS automata the chip runs at once: needs a library built with -DCORTO_TOPO_TIMES (tools/ab_build.sh times CORTO_TOPO_TIMES; run with
CORTO_HIP_LIB_PATH=corto_amd/lib_times/libcorto_hip.so).  One batch of $NB blobs ($MESH as tools/kt_probe_irregular.py): per workgroup the shader clock at entry and exit
and where it ran -> concurrency over time, residency per CU / SIMD, start-time spread."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
kind = os.environ.get("MESH", "delaunay"); nb = int(os.environ.get("NB", "2048"))
gen = {"delaunay": lambda i: synth.delaunay_disc(2310, seed=i, holes=6 + i % 5), "flipped": lambda i: synth.bumpy_sphere_flipped(64, 32, seed=i), "regular": lambda i: synth.bumpy_sphere(64, 32, seed=i)}[kind]
blobs = [ca.encode(gen(i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(nb)]
ctx = ca.Context(0); ctx.set_profiling(True)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
for i in range(3): b.decode(); b.sync()
print({k: round(v["ms"], 4) for k, v in b.kernel_times().items()})
out = np.zeros(8 * 8192, dtype=np.uint32)
L = ca.lib(); L.crthip_debug_topo_times.argtypes = [C.c_void_p]
print("rc", L.crthip_debug_topo_times(out.ctypes.data_as(C.c_void_p)))
o = out.reshape(8192, 8)[:nb].astype(np.uint64)
t0 = o[:, 0] | (o[:, 1] << np.uint64(32)); t1 = o[:, 2] | (o[:, 3] << np.uint64(32))
base = t0.min(); t0 = (t0 - base).astype(np.float64); t1 = (t1 - base).astype(np.float64)
hw = o[:, 4].astype(np.int64); xcc = o[:, 5].astype(np.int64) & 0xF
simd = (hw >> 4) & 3; cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
dur = t1 - t0
print("workgroups", nb, "span clocks %.0f" % t1.max(), "duration mean %.0f max %.0f" % (dur.mean(), dur.max()), "start spread: p50 %.0f p90 %.0f max %.0f" % (np.percentile(t0, 50), np.percentile(t0, 90), t0.max()))
ev = sorted([(t, 1) for t in t0] + [(t, -1) for t in t1]); c = 0; last = 0; area = {}
for t, d in ev:
    area[c] = area.get(c, 0) + (t - last); last = t; c += d
tot = sum(area.values())
print("mean concurrent workgroups %.1f; time share by concurrency (buckets of 256):" % (sum(k * v for k, v in area.items()) / tot), {k: round(sum(v for c2, v in area.items() if c2 // 256 == k) / tot, 3) for k in sorted(set(c2 // 256 for c2 in area))})
ncu = len(set(cu.tolist())); per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
print("distinct CUs", ncu, "workgroups per CU min/mean/max", per_cu.min(), round(per_cu.mean(), 2), per_cu.max(), "SIMD histogram", np.bincount(simd, minlength=4).tolist())
# peak residency per CU: max overlapping workgroups on the same CU
peak = []
for u in np.unique(cu)[:64]:
    m = cu == u; e2 = sorted([(t, 1) for t in t0[m]] + [(t, -1) for t in t1[m]]); c = 0; pk = 0
    for t, d in e2: c += d; pk = max(pk, c)
    peak.append(pk)
print("peak resident workgroups on a CU (first 64 CUs): min %d max %d mean %.2f" % (min(peak), max(peak), np.mean(peak)))
