"""Which probability tables does the device's dictionary builder get wrong (against the oracle)?  Prints them.  python tools/tun_batch_probe.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import corto_amd as ca
from oracle import oracle as oc
from test_gpu_parity import _run_blocks
ctx = ca.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
tables = []
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    n = int(rng.integers(2, 10))
    kind = t % 6
    if kind == 0: p = np.sort(rng.integers(0, 256, n))[::-1]
    elif kind == 1: p = np.sort((255 * rng.dirichlet(np.ones(n) * 0.3)).astype(int))[::-1]
    elif kind == 2: p = np.array([max(250 - n, 1)] + list(np.sort(rng.integers(0, 4, n - 1))[::-1]))
    elif kind == 3: p = np.sort((255 * rng.dirichlet(np.ones(n) * 4)).astype(int))[::-1]
    elif kind == 4: p = np.full(n, max(255 // n, 1))
    else: p = np.sort(rng.integers(0, 3, n))[::-1]
    tables.append(np.stack([rng.permutation(256)[:n], np.clip(p, 0, 255)], 1).astype(np.uint8))
blocks, sizes, expect, keep = [], [], [], []
for pr in tables:
    idx, ln, tab = oc.tunstall_tables(pr)
    idx, ln = np.asarray(idx).astype(int), np.asarray(ln).astype(int)
    if (ln > 0).sum() < 256: continue
    words = np.concatenate([np.asarray(tab)[idx[c]:idx[c] + ln[c]] for c in range(256)])
    hdr = bytes([len(pr)]) + pr.tobytes() + int(len(words)).to_bytes(4, "little") + (256).to_bytes(4, "little")
    blocks.append(np.frombuffer(hdr + bytes(range(256)), dtype=np.uint8)); sizes.append(len(words)); expect.append((words, ln)); keep.append(pr)
outs, _ = _run_blocks(ctx, blocks, sizes)
nbad = 0
for pr, o, (e, ln) in zip(keep, outs, expect):
    if not np.array_equal(o, e):
        nbad += 1
        if nbad <= 12:
            d = int(np.argmax(o != e)); c = int(np.searchsorted(np.cumsum(ln), d, side="right"))
            print("BAD n", len(pr), "probs", pr[:, 1].tolist(), "first differing byte", d, "in word", c)
print("tables", len(keep), "bad", nbad)
