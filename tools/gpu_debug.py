"""Ad-hoc GPU bring-up script (not a test): decode each golden fixture, print the first mismatch per array."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import corto_amd as ca
from conftest import ALL_CASES, load_golden

ctx = ca.Context(0)
ctx.set_profiling(True)
for name in ALL_CASES if len(sys.argv) < 2 else sys.argv[1:]:
    g = load_golden(name)
    cc = int(g["color_components"])
    b = ca.Batch(ctx, [g["crt"]])
    b.allocate_outputs(color_components=cc if "color" in g else None, fill=0)
    t = time.time(); b.decode(); st = b.sync(raise_on_error=False); dt = time.time() - t
    got = b.host_outputs(0)
    msg = []
    if "index" in g:
        cl = b.debug_read(0, "clers", len(g["_clers"]) + 16)
        msg.append("clers %s" % ("ok" if np.array_equal(cl, g["_clers"]) else "BAD"))
        pr = b.debug_read(0, "prediction", g["_prediction"].size * 4).view(np.uint32).reshape(-1, 3)
        msg.append("pred %s" % ("ok" if np.array_equal(pr[1:], g["_prediction"][1:]) else "BAD"))
    for k in ("index", "position", "uv", "color", "radius", "normal"):
        if k in g:
            a, e = got[k], g[k]
            if a.tobytes() == e.tobytes():
                msg.append(k + " ok")
            else:
                bad = np.argwhere(a.reshape(len(a), -1) != e.reshape(len(e), -1))
                msg.append("%s BAD n=%d first=%s got=%s exp=%s" % (k, len(bad), bad[0], a.reshape(len(a), -1)[bad[0][0]], e.reshape(len(e), -1)[bad[0][0]]))
    print("%-20s status %s %.1f ms | %s" % (name, st, dt * 1e3, " | ".join(msg)))
    kt = b.kernel_times()
    print("    ", {k: round(v["ms"], 3) for k, v in kt.items()})
