"""Robustness probe (run under `timeout`): byte-flipped copies of the golden fixtures, decoded in one batch next to intact ones.
A corrupt blob may fail (status != 0) or decode garbage; it must not fault, hang, or disturb its neighbours."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import corto_amd as ca
from conftest import MESH_CASES, CLOUD_CASES, load_golden, aligned
from oracle import oracle as oc
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = ca.Context(0)
names = list(MESH_CASES) + list(CLOUD_CASES)
nmut = int(os.environ.get("NMUT", "64"))
good = load_golden("c4_unit")
blobs, kinds = [], []
for i in range(nmut):
    g = load_golden(names[i % len(names)])
    b = g["crt"].copy()
    h = oc.parse_header(b)
    body = h["body_offset"]
    mode = i % 4
    if mode == 0:   # a few random byte flips anywhere in the body
        for p in rng.integers(body, len(b), 6): b[p] ^= rng.integers(1, 256)
    elif mode == 1: # a burst
        p = int(rng.integers(body, max(body + 1, len(b) - 64))); b[p:p + 48] ^= 0xA5
    elif mode == 2: # random garbage tail
        p = int(rng.integers(body, len(b))); b[p:] = rng.integers(0, 256, len(b) - p, dtype=np.uint8)
    else:           # zeroed window
        p = int(rng.integers(body, max(body + 1, len(b) - 200))); b[p:p + 160] = 0
    blobs.append(aligned(b)); kinds.append(mode)
    if i % 8 == 7:
        blobs.append(aligned(good["crt"])); kinds.append(-1)
accepted = []
for i, b in enumerate(blobs):            # the host walk rejects what it can prove truncated/inconsistent
    try:
        ca.probe(b); ca.Batch(ctx, [b]).close(); accepted.append(i)
    except ca.CortoError:
        pass
print("host walk accepted %d of %d" % (len(accepted), len(blobs)), flush=True)
bt = ca.Batch(ctx, [blobs[i] for i in accepted])
bt.allocate_outputs(fill=0)
bt.decode()
st = bt.sync(raise_on_error=False)
print("device status histogram:", {int(k): int((st == k).sum()) for k in np.unique(st)}, flush=True)
ref = oc.decode(good["crt"])
for j, i in enumerate(accepted):
    if kinds[i] == -1:
        assert st[j] == 0
        got = bt.host_outputs(j)
        for k in ("position", "normal", "color", "uv", "index"):
            assert got[k].tobytes() == ref[k].tobytes(), (j, k)
print("intact neighbours bit-exact; fuzz probe ok", flush=True)
