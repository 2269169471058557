"""Per-kernel HIP-event times of one batch of 256 irregular blobs, unpipelined: $MESH = flipped (bumpy_sphere_flipped, $FLIP sets the
flip probability) | torus | holey | strip | grid128 | delaunay (discs with holes: bench.py's `realistic` blobs) | icosphere | decimated | confetti | cone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
flip = float(os.environ.get("FLIP", "0.5"))
kind = os.environ.get("MESH", "flipped")                      # flipped | torus | holey | strip | grid128
gen = {"flipped": lambda i: synth.bumpy_sphere_flipped(64, 32, seed=i, flip=flip), "torus": lambda i: synth.torus(48, 24, seed=i),
       "holey": lambda i: synth.holey_disc(40, seed=i), "strip": lambda i: synth.strip(400, seed=i), "grid128": lambda i: synth.bumpy_sphere(128, 64, seed=i),
       "delaunay": lambda i: synth.delaunay_disc(2310, seed=i, holes=6 + i % 5), "delaunay0": lambda i: synth.delaunay_disc(2310, seed=i, holes=0),
       "icosphere": lambda i: synth.icosphere(4, seed=i), "decimated": lambda i: synth.decimated(synth.icosphere(4, seed=i), keep=0.8, seed=i),
       "confetti": lambda i: synth.confetti(600, seed=i), "cone": lambda i: synth.cone_fan(128, 16, seed=i)}[kind]
blobs = [ca.encode(gen(i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(int(os.environ.get('NB', '256')))]
ctx = ca.Context(0); ctx.set_profiling(True)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
acc = {}
N = 8
for i in range(N + 3):
    b.decode(); b.sync()
    if i >= 3:
        for k, v in b.kernel_times().items():
            a = acc.setdefault(k, [0.0, 0]); a[0] += v["ms"]; a[1] += v.get("launches", 1)
st = b.stats()
cl = np.bincount(np.concatenate([__import__("oracle.oracle", fromlist=["x"]).decode(ca.aligned_blob(x), trace=True)["_clers"] for x in blobs[:8]]), minlength=7)
t_all = __import__('time').perf_counter()
for _ in range(5): b.decode(); b.sync()
print('NB', len(blobs), 'wall ms per decode', round((__import__('time').perf_counter() - t_all) / 5 * 1e3, 4))
print("symbols of 8 blobs V L R E B D S", cl.tolist(), "nvert", ca.probe(blobs[0]).nvert, "nface", ca.probe(blobs[0]).nface)
print(kind, "flip", flip, {k: (round(v[0] / N, 4), v[1] // N) for k, v in acc.items()}, "dicts", st.tunstall_dictionaries, "of", st.tunstall_streams, "fallbacks", st.topology_fallbacks, "scale", st.topology_scale, "delta walked", st.delta_walked, "redone", st.delta_redone, "clers", st.clers_symbols // 256)
