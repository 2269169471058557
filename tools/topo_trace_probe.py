"""What each kind of step of the CLERS automaton costs, measured: a -DCORTO_TOPO_STAMPS build (tools/ab_build.sh stamps CORTO_TOPO_STAMPS) leaves the
shader clock of every dispatch in a trace (k_mesh.hip: TOPO_ASM_STAMP); `gpu` dumps the traces of the first 16 blobs of a few families
(-> gpurun_out/topo_trace.npz), `fit` (CPU) labels every interval with what the host model (tools/topo_run_model.py) did in it.
Small blobs only (the trace sits 32 KB up in the blob's LDS).  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def families():
    from corto_amd import synth
    return {
        "reg": (lambda i: synth.bumpy_sphere(64, 32, seed=i, color_components=4), 64),
        "reg32": (lambda i: synth.bumpy_sphere(32, 16, seed=i, color_components=4), 32),
        "reg96": (lambda i: synth.bumpy_sphere(96, 20, seed=i, color_components=4), 32),
        "f50": (lambda i: synth.bumpy_sphere_flipped(64, 32, seed=i, flip=0.5), 64),
        "f10": (lambda i: synth.bumpy_sphere_flipped(64, 32, seed=i, flip=0.1), 64),
        "f25s": (lambda i: synth.bumpy_sphere_flipped(40, 20, seed=i, flip=0.25), 32),
        "closed": (lambda i: synth.closed_sphere(40 + i % 8, 20 + i % 5, seed=i), 16),
        "torus": (lambda i: synth.torus(40 + 2*(i % 4), 20 + i % 3, seed=i), 12),
        "disc": (lambda i: synth.holey_disc(30 + i % 6, seed=i), 12),
        "strip": (lambda i: synth.strip(300 + 20*(i % 4), seed=i), 8),
    }

def blobs_of(name):
    import corto_amd as ca
    gen, n = families()[name]
    return [ca.encode(gen(i), position_bits=14, uv_bits=12, normal_bits=10) for i in range(n)]


FAMS = ["reg", "f50", "f10", "closed", "strip", "torus"]   # (not the disc: its front outgrows 32 KB once the context has learnt its scale)

def gpu():
    import ctypes as C
    import corto_amd as ca
    L = ca.lib(); L.crthip_debug_topo_trace.argtypes = [C.c_void_p]
    out = {}
    for name in FAMS:
        blobs = blobs_of(name)[:16]
        ctx = ca.Context(0)
        b = ca.Batch(ctx, blobs, device_arena=ca.upload_arena(blobs, 0)); b.allocate_outputs()
        for i in range(3): b.decode(); b.sync()
        o = np.zeros(16*8192, dtype=np.uint32)
        assert L.crthip_debug_topo_trace(o.ctypes.data_as(C.c_void_p)) == 0
        out[name] = o.reshape(16, 8192)[:len(blobs)].copy()
        print(name, "stamps per blob", (out[name][:, :8191] != 0).sum(axis=1)[:4], "fallbacks", b.stats().topology_fallbacks if hasattr(b.stats(), "topology_fallbacks") else "?", flush=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "topo_trace.npz"), **out)

def fit():
    import corto_amd as ca
    import topo_run_model as tm
    from oracle import oracle as oc
    from collections import defaultdict
    z = np.load(os.path.join(ROOT, "gpurun_out", "topo_trace.npz"))
    for name in FAMS:
        if name not in z: continue
        agg = defaultdict(lambda: [0, 0.0, 0])
        total = 0.0
        for bi, blob in enumerate(blobs_of(name)[:len(z[name])]):
            blob = ca.aligned_blob(blob); r = oc.decode(blob, trace=True)
            m = tm.Model(list(r["_clers"]), r["nvert"], r["nface"], ca.probe_groups(blob), ref_faces=r["index"]); m.run()
            tr = z[name][bi]; n = len(r["_clers"])
            at = [i for i in range(min(n + 1, 8191)) if tr[i]]
            if len(at) < 2: continue
            ends = at[1:] + [n + 1]
            tend = [int(tr[j]) if j <= n and j < 8191 and tr[j] else int(tr[8191]) for j in ends]
            ev = sorted(m.events, key=lambda e: e[0]); k = 0
            for i, j, t1 in list(zip(at, ends, tend))[:-1]:          # (the last interval holds the trace's own copy-out)
                dt = (t1 - int(tr[i])) & 0xFFFFFFFF
                kinds = []; nsym = 0; dead = 0
                while k < len(ev) and ev[k][0] < i: k += 1
                kk = k
                while kk < len(ev) and ev[kk][0] < j:
                    e = ev[kk]
                    if e[1] == 'dead': dead += 1
                    elif e[1] == 'pop' and e[0] == i and not kinds: pass      # (the pop that led to this dispatch belongs to the interval before)
                    else: kinds.append(e[1] if e[1] not in ('run', 'mix', 'ends') else e[1]); nsym += e[2] if e[1] in ('run', 'mix', 'ends') else 0
                    kk += 1
                # the pop after a chain end sits at symbol index j (the next dispatch): take it and its dead entries
                k2 = kk
                while k2 < len(ev) and ev[k2][0] == j and ev[k2][1] in ('dead', 'pop', 'dpop'):
                    if ev[k2][1] == 'dead': dead += 1
                    else: kinds.append(ev[k2][1])
                    k2 += 1
                    if ev[k2 - 1][1] != 'dead': break
                label = "+".join(kinds) + ("" if not dead else " dead<=64" if dead <= 64 else " dead>64")
                a = agg[label]; a[0] += 1; a[1] += dt; a[2] += nsym
                total += dt
        print("== %s: %.0f clocks a blob in the trace" % (name, total/len(z[name])))
        for label, (cnt, clk, nsym) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
            print("   %-34s x %6.1f a blob  %7.0f clocks each  %5.1f %%%s" % (label, cnt/len(z[name]), clk/cnt, 100*clk/total, "  (%.1f symbols each)" % (nsym/cnt) if nsym else ""))

if __name__ == "__main__":
    {"gpu": gpu, "fit": fit}[sys.argv[1]]()
