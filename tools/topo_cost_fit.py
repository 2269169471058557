"""What each KIND of step of the CLERS automaton costs (k_topology_lds), by regression: per-blob shader clocks from a -DCORTO_TOPO_STAMPS build
(`gpu` mode, on the GPU box -> gpurun_out/topo_cost.npz) against the per-blob step counts of the host model (tools/topo_run_model.py; `fit`
mode, on the CPU).  Development aid: decides which step is worth folding into which."""
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

def gpu():
    import ctypes as C
    import corto_amd as ca
    L = ca.lib(); L.crthip_debug_topo_stamps.argtypes = [C.c_void_p]
    out = {}
    for name in families():
        blobs = blobs_of(name)
        ctx = ca.Context(0)
        b = ca.Batch(ctx, blobs, device_arena=ca.upload_arena(blobs, 0)); b.allocate_outputs()
        runs = []
        for i in range(4):
            b.decode(); b.sync()
            o = np.zeros(48*4096, dtype=np.uint32)
            assert L.crthip_debug_topo_stamps(o.ctypes.data_as(C.c_void_p)) == 0
            runs.append(o.reshape(4096, 48)[:len(blobs)].copy())
        r = np.stack(runs)
        out[name] = r[:, :, 15].min(axis=0)                      # best of four decodes: the first one pays cold instruction fetches
        out[name + "_err"] = r[-1, :, 25]
        out[name + "_phases"] = r[-1, :, :24]
        ph = r[-1].astype(np.float64).mean(axis=0)
        print(name, len(blobs), "clocks mean %.0f" % out[name].mean(), "fallbacks", int((r[-1, :, 25] == 2).sum()),
              "| ISA block: %.0f clocks, %.1f entries, %.0f symbols | C++ symbol: %.0f clocks, %.1f steps | gate fetch (C++): %.0f clocks, %.1f | prologue %.0f" %
              (ph[0], ph[8], ph[16], ph[3], ph[11], ph[4], ph[12], ph[5]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "topo_cost.npz"), **out)

FEATURES = ["const", "runs", "run_pairs", "mixes", "mix_symbols", "ends", "end_symbols", "pops", "dead64", "dpops", "seeds", "V", "Vlone", "L", "R", "E", "B", "D", "S"]
def counts(blob):
    import corto_amd as ca
    import topo_run_model as tm
    from oracle import oracle as oc
    blob = ca.aligned_blob(blob)
    r = oc.decode(blob, trace=True)
    m = tm.Model(list(r["_clers"]), r["nvert"], r["nface"], ca.probe_groups(blob), ref_faces=r["index"])
    m.run(); s = m.stats
    f = dict(const=1, runs=s["runs"], run_pairs=s["run_pairs"], mixes=s["mixes"], mix_symbols=s["mix_symbols"], ends=s["ends"], end_symbols=s["end_symbols"],
             pops=s["pops"], dead64=s["dead"]/64.0, dpops=s["dpops"], seeds=s["seeds"], Vlone=s["lone_v_before_run"])
    for k in "VLREBDS": f[k] = s["ser"][k]
    f["V"] -= f["Vlone"]
    return [f[k] for k in FEATURES], len(r["_clers"])

def fit():
    z = np.load(os.path.join(ROOT, "gpurun_out", "topo_cost.npz"))
    X, y, fam, ns = [], [], [], []
    for name in families():
        clk = z[name]; err = z[name + "_err"]
        for i, blob in enumerate(blobs_of(name)):
            if err[i]: continue
            c, n = counts(blob)
            X.append(c); y.append(float(clk[i])); fam.append(name); ns.append(n)
        print(name, "done", flush=True)
    X = np.array(X, dtype=np.float64); y = np.array(y)
    from scipy.optimize import nnls
    w, _ = nnls(X, y)
    pred = X @ w
    print("clocks per unit:")
    for k, v in zip(FEATURES, w): print("  %-12s %8.1f" % (k, v))
    fam = np.array(fam)
    for name in families():
        s = fam == name
        if not s.any(): continue
        share = (X[s]*w).mean(axis=0)
        print("%-7s measured %8.0f  fitted %8.0f  (rms err %.1f %%)  shares: %s" % (name, y[s].mean(), pred[s].mean(), 100*np.sqrt(np.mean(((pred[s] - y[s])/y[s])**2)),
              " ".join("%s %.0f%%" % (k, 100*v/pred[s].mean()) for k, v in zip(FEATURES, share) if v/pred[s].mean() >= 0.02)))

if __name__ == "__main__":
    {"gpu": gpu, "fit": fit}[sys.argv[1]]()
