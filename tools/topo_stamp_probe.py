"""Where the CLERS automaton's time goes (k_topology_lds): needs the library built with CORTO_BUILD_DEFINES=CORTO_TOPO_STAMPS
(python -m corto_amd.build --force).  Shader clocks, entries and symbols per phase (the ISA block with its run and mix steps, the C++ symbols, the gate fetch), averaged over the 256 blobs of a batch;
$FLIP > 0 takes the irregular blobs (bumpy_sphere_flipped)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
flip = float(os.environ.get("FLIP", "0"))
gen = (lambda i: synth.bumpy_sphere_flipped(64, 32, seed=i, flip=flip)) if flip > 0 else (lambda i: synth.bumpy_sphere(64, 32, seed=i, color_components=4))
blobs = [ca.encode(gen(i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(256)]
ctx = ca.Context(0); ctx.set_profiling(True)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
for i in range(4): b.decode(); b.sync()
print({k: round(v["ms"], 4) for k, v in b.kernel_times().items()})
out = np.zeros(48*4096, dtype=np.uint32)
L = ca.lib(); L.crthip_debug_topo_stamps.argtypes = [C.c_void_p]
print("rc", L.crthip_debug_topo_stamps(out.ctypes.data_as(C.c_void_p)))
o = out.reshape(4096, 48)[:256].astype(np.float64)
names = ["ISA block", None, None, "C++ symbol", "gate fetch", "prologue"]   # (1, 2: the run and mix steps when they were asm statements of their own; sections of the block now)
tot = o[:, 15]
print("flip %.2f  total clocks: mean %.0f  max %.0f  (blob %d)" % (flip, tot.mean(), tot.max(), int(tot.argmax())))
for i, n in enumerate(names):
    if n is None: continue
    clk, cnt, sym = o[:, i].mean(), o[:, 8 + i].mean(), o[:, 16 + i].mean()
    print("  %-11s clocks %8.0f (%4.1f %%)  steps %7.1f  symbols %7.1f  clocks/step %7.1f  clocks/symbol %7.1f" % (n, clk, 100*clk/tot.mean(), cnt, sym, clk/max(cnt, 1), clk/max(sym, 1)))
print("  unaccounted %.1f %%" % (100*(1 - o[:, :6].sum(axis=1).mean()/tot.mean())))
w = int(tot.argmax())
print("  slowest blob:", {n: (int(o[w, i]), int(o[w, 8 + i]), int(o[w, 16 + i])) for i, n in enumerate(names) if n})
