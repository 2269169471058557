"""Probe: which synthetic meshes outgrow the LDS automaton's ring / pool and are redone on the HBM front (stats.topology_fallbacks),
and that every one of them decodes bit-exact either way."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
import corto_amd as ca
from corto_amd import synth
from oracle import oracle as oc
ctx = ca.Context(0)
cases = [("strip400", synth.strip(400, seed=3)), ("sphere24", synth.bumpy_sphere(24, 12, seed=5)), ("torus48x24", synth.torus(48,24,seed=1)), ("torus100x50", synth.torus(100,50,seed=1)),
         ("torus200x100", synth.torus(200,100,seed=2)), ("disc40", synth.holey_disc(40, seed=2, color_components=4)), ("sphere256", synth.bumpy_sphere(256,125,seed=1))]
for name, m in cases:
    blob = ca.encode(m, normal_prediction=ca.BORDER)
    b = ca.Batch(ctx, [blob]); b.allocate_outputs(fill=0); b.decode(); st = b.sync(raise_on_error=False)
    exp = oc.decode(blob)
    ok = all(b.host_outputs(0)[k].tobytes() == exp[k].tobytes() for k in ("position","normal","color","uv","index"))
    print(name, m.nvert, m.nface, "status", st[0], "fallbacks", b.stats().topology_fallbacks, "bit-exact", ok, flush=True)
