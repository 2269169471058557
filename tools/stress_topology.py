"""Random meshes of every synthetic family (sizes, flip rates, hole fractions, random group cuts, shuffles, merges) through the GPU decoder against the
oracle, byte for byte, and the HBM-redo counter beside it: the automaton's wave-wide steps (run, mix, chain ends) are hand-written ISA, a
wrong front often only runs out of slots, falls back and still decodes right - so mismatches AND unexpected fallbacks are what to look at.
    python tools/stress_topology.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
from oracle import oracle as oc
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ca.Context(0)
KEYS = ("index", "position", "normal", "uv", "color")
total = bad = 0
fallbacks = []
for rd in range(rounds):
    meshes = []; kinds = []
    for _ in range(24):
        k = int(rng.integers(0, 17)); s = int(rng.integers(0, 1 << 30))
        if k == 0: m = synth.bumpy_sphere(int(rng.integers(8, 90)), int(rng.integers(4, 45)), seed=s)
        elif k == 1: m = synth.bumpy_sphere_flipped(int(rng.integers(8, 90)), int(rng.integers(4, 45)), seed=s, flip=float(rng.choice([0.01, 0.05, 0.2, 0.5, 0.9, 1.0])))
        elif k == 2: m = synth.holey_disc(int(rng.integers(8, 48)), seed=s, hole_frac=float(rng.uniform(0.02, 0.3)))
        elif k == 3: m = synth.torus(int(rng.integers(6, 60)), int(rng.integers(4, 30)), seed=s)
        elif k == 4: m = synth.strip(int(rng.integers(10, 500)), seed=s)
        elif k == 5: m = synth.closed_sphere(int(rng.integers(6, 40)), int(rng.integers(4, 20)), seed=s)
        elif k == 6: m = synth.shuffled(synth.bumpy_sphere_flipped(int(rng.integers(8, 40)), int(rng.integers(4, 20)), seed=s, flip=0.3))
        elif k == 7: m = synth.merge([synth.bumpy_sphere(int(rng.integers(6, 24)), int(rng.integers(4, 12)), seed=s), synth.holey_disc(int(rng.integers(8, 20)), seed=s + 1, color_components=4), synth.torus(12, 6, seed=s + 2)])
        # non-lattice connectivity (round 5): icosphere, Delaunay with holes, high-valence cones, decimated spheres, confetti - plain, shuffled and merged
        elif k == 8: m = synth.icosphere(int(rng.integers(0, 4)), seed=s)
        elif k == 9: m = synth.delaunay_disc(int(rng.integers(30, 2400)), seed=s, holes=int(rng.integers(0, 14)), drop=float(rng.uniform(0.0, 0.5)))
        elif k == 10: m = synth.cone_fan(int(rng.integers(5, 260)), int(rng.integers(1, 6)), seed=s, closed=bool(rng.integers(0, 2)), flip=float(rng.uniform(0, 1)))
        elif k == 11: m = synth.decimated(synth.icosphere(int(rng.integers(1, 4)), seed=s), keep=float(rng.uniform(0.15, 0.9)), seed=s, max_valence=int(rng.integers(8, 40)))
        elif k == 12: m = synth.confetti(int(rng.integers(10, 420)), seed=s, max_faces=int(rng.integers(1, 13)))
        elif k == 13: m = synth.shuffled(synth.delaunay_disc(int(rng.integers(30, 900)), seed=s, holes=int(rng.integers(0, 9))), seed=s, faces=bool(rng.integers(0, 2)), verts=bool(rng.integers(0, 2)))
        # non-manifold input (round 6): fins, duplicated / reversed faces, bow-ties, glued pairs on bases of three kinds (encoder.cpp:450-504,633-636)
        elif k == 15: m = synth.non_manifold([synth.delaunay_disc(int(rng.integers(30, 1500)), seed=s, holes=int(rng.integers(0, 8))), synth.bumpy_sphere_flipped(int(rng.integers(8, 50)), int(rng.integers(4, 25)), seed=s, flip=0.4), synth.icosphere(int(rng.integers(1, 4)), seed=s)][s % 3],
                                             seed=s, fins=int(rng.integers(0, 40)), dups=int(rng.integers(0, 30)), reversed_dups=int(rng.integers(0, 30)), bowties=int(rng.integers(0, 12)), glue=int(rng.integers(0, 10)), shuffle_faces=bool(rng.integers(0, 2)))
        elif k == 16: m = synth.non_manifold(synth.merge([synth.confetti(int(rng.integers(5, 80)), seed=s), synth.cone_fan(int(rng.integers(5, 100)), 2, seed=s + 1), synth.torus(int(rng.integers(6, 20)), int(rng.integers(4, 10)), seed=s + 2)]), seed=s, fins=int(rng.integers(0, 60)), dups=int(rng.integers(0, 60)), reversed_dups=int(rng.integers(0, 20)), bowties=int(rng.integers(0, 20)), glue=0)
        else: m = synth.shuffled(synth.merge([synth.decimated(synth.icosphere(2, seed=s), keep=0.5, seed=s), synth.confetti(int(rng.integers(5, 60)), seed=s + 1), synth.cone_fan(int(rng.integers(64, 130)), 2, seed=s + 2),
                                              synth.delaunay_disc(int(rng.integers(40, 300)), seed=s + 3, holes=3)]), seed=s)
        if rng.random() < 0.3 and m.nface > 8:           # several groups (each starts from an empty front): random cuts
            cuts = sorted(set(int(c) for c in rng.integers(1, m.nface, int(rng.integers(1, 4)))))
            m.groups = cuts + [m.nface]
        meshes.append(m); kinds.append(k)
    if rd % 5 == 4:                                      # now and then something beyond the LDS symbol window (slides) and the 64-entry scans
        meshes[0] = synth.bumpy_sphere_flipped(int(rng.integers(100, 180)), int(rng.integers(50, 90)), seed=rd, flip=float(rng.choice([0.02, 0.5])))
        meshes[1] = synth.holey_disc(int(rng.integers(60, 100)), seed=rd, hole_frac=0.1)
        kinds[0], kinds[1] = 1, 2
    blobs = [ca.aligned_blob(ca.encode(m, position_bits=int(rng.integers(10, 18)), uv_bits=12, normal_bits=10,
                                       normal_prediction=[ca.BORDER if kinds[i] != 15 else ca.DIFF, ca.ESTIMATED, ca.DIFF][i % 3])) for i, m in enumerate(meshes)]   # (family 15's back-to-back pairs make BORDER copy a 0/0 estimate to the output: out of contract as upstream)
    u16 = bool(rd & 1)
    refs = [oc.decode(blob, index16=u16, color_components=4) for blob in blobs]
    for attempt in range(2):                        # the second pass runs with the slots the first one taught the context
        b = ca.Batch(ctx, blobs); b.allocate_outputs(fill=0, index16=u16, color_components=4); b.decode()
        st = b.sync()
        assert (np.asarray(st) == 0).all(), ("status", rd, st)
        for i, blob in enumerate(blobs):
            r = refs[i]
            got = b.host_outputs(i)
            for key in KEYS:
                if key in r and got[key].tobytes() != r[key].tobytes():
                    bad += 1
                    g_, r_ = got[key].reshape(len(got[key]), -1), r[key].reshape(len(r[key]), -1)
                    rows = np.flatnonzero((g_.view(np.uint8).reshape(len(g_), -1) != r_.view(np.uint8).reshape(len(r_), -1)).any(1))
                    print("MISMATCH round", rd, "blob", i, key, "nface", r["nface"], "nvert", r["nvert"], "attempt", attempt, "rows", len(rows), "first", rows[:4].tolist(), "last", rows[-2:].tolist())
            total += 1
        fallbacks.append(int(b.stats().topology_fallbacks))
        b.close()
        if attempt == 1 and fallbacks[-1]:             # who still falls back with the slots learnt: each blob alone on a fresh context, three times (scale 1, 4, 16)
            for i, blob in enumerate(blobs):
                c1 = ca.Context(0); n = []
                for _ in range(3):
                    b1 = ca.Batch(c1, [blob]); b1.allocate_outputs(fill=0); b1.decode(); b1.sync(); n.append(int(b1.stats().topology_fallbacks)); b1.close()
                c1.close()
                if n[2]:
                    # a front of more than 4 096 + 4 096 records has no LDS form - the HBM front is its path by design: every BOUNDARY edge
                    # stays in the pool for good, every DELAYed one until it is popped, and the queue of a mesh with handles is ten times a sphere's
                    cl = oc.decode(blob, trace=True)["_clers"]
                    nb, nd = int((cl == 4).sum()), int((cl == 5).sum())
                    big = nd > 3600 or meshes[i].nface > 60000                    # (round 6: the BOUNDARY edges share one slot - the DELAYed ones alone need the pool)
                    print("  fallback by capacity:" if big else "  persistent fallback:", "round", rd, "blob", i, "kind", kinds[i], "nface", meshes[i].nface, "nvert", meshes[i].nvert, "BOUNDARY", nb, "DELAY", nd, n)
print("decodes", total, "mismatching arrays", bad, "fallbacks per batch (first pass, second pass ...)", fallbacks)
