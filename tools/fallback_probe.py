"""Which blobs keep falling back to the HBM front, and from what size on: each mesh alone on a fresh context, three decodes (the context learns after the first).
    python tools/fallback_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
from corto_amd import synth
cases = [("strip%d" % n, synth.strip(n, seed=1)) for n in (50, 100, 120, 130, 200, 300, 377, 430)]
cases += [("disc%d/%.2f" % (n, h), synth.holey_disc(n, seed=3, hole_frac=h)) for n, h in ((30, 0.2), (48, 0.2), (48, 0.05), (80, 0.1), (100, 0.1))]
cases += [("confetti%d" % n, synth.confetti(n, seed=2)) for n in (100, 240, 400)]
cases += [("torus%dx%d" % (a, b), synth.torus(a, b, seed=2)) for a, b in ((40, 20), (60, 30), (100, 50))]
def cut(m, cuts):
    m.groups = list(cuts) + [m.nface]; return m
cases += [("strip377 g%s" % (g,), cut(synth.strip(377, seed=5), g)) for g in ((100,), (376,), (377,), (10, 700), (300, 301, 302))]
cases += [("strip430 s%d" % sd, synth.strip(430, seed=sd)) for sd in (7, 12345, 99999999)]
cases += [("disc80 few holes", synth.holey_disc(80, seed=3, hole_frac=0.01)), ("disc100 no holes", synth.holey_disc(100, seed=3, hole_frac=0.0))]
for name, m in cases:
    blob = ca.aligned_blob(ca.encode(m))
    c = ca.Context(0); out = []
    for _ in range(3):
        b = ca.Batch(c, [blob]); b.allocate_outputs(fill=0); b.decode(); b.sync(); s = b.stats(); out.append((int(s.topology_fallbacks), int(s.topology_scale))); b.close()
    c.close()
    print("%-14s nvert %6d nface %6d 2V-F %6d  (fallbacks, scale) x3: %s" % (name, m.nvert, m.nface, 2 * m.nvert - m.nface, out))
