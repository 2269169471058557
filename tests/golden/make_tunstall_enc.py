#!/usr/bin/env python3
"""Known-answer vectors for the GPU Tunstall ENCODER stage, made FROM THE UNMODIFIED REFERENCE
(oracle/_ref/libcorto_ref.so: OutStream::tunstall_compress, src/cstream.cpp:89-109).  Run in the build container:

    python tests/golden/make_tunstall_enc.py

Writes tests/golden/tunstall_enc_kat.npz: input_XX (symbols) and block_XX (the bytes the reference appended to its stream).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refcodec as rc      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def streams():
    rng = np.random.default_rng(2026)
    out = []
    # bit-width logs as the attribute coder makes them: a few neighbouring values, one dominant
    for n, centre, spread in ((2112, 4, 1.2), (4319, 2, 0.8), (700, 9, 2.5), (65, 3, 1.0), (9001, 6, 0.6)):
        out.append(np.clip(np.rint(rng.normal(centre, spread, n)), 0, 31).astype(np.uint8))
    # CLERS-like: six symbols, very uneven
    out.append(rng.choice(np.arange(6, dtype=np.uint8), 5000, p=[0.55, 0.2, 0.12, 0.08, 0.03, 0.02]))
    # low entropy: the run-of-the-likeliest-symbol seed (count >= 16), words of tens of symbols
    out.append((rng.random(6000) < 0.02).astype(np.uint8) * 7)
    out.append(np.where(rng.random(3000) < 0.004, 3, 200).astype(np.uint8))
    z = np.zeros(2500, dtype=np.uint8); z[rng.integers(0, 2500, 9)] = rng.integers(1, 5, 9); out.append(z)
    # flat: many symbols, short words
    out.append(rng.integers(0, 64, 3000).astype(np.uint8))
    out.append(rng.integers(0, 200, 1500).astype(np.uint8))
    # edges: tiny, one symbol, stream ending inside a word, odd/even lengths
    out.append(np.array([5], dtype=np.uint8))
    out.append(np.array([5, 5, 5, 5], dtype=np.uint8))
    out.append(np.array([1, 2], dtype=np.uint8))
    out.append(np.array([1, 2, 1], dtype=np.uint8))
    out.append(np.array([0] * 40 + [1], dtype=np.uint8))
    out.append(np.array([0] * 41, dtype=np.uint8)[:41] | np.array([0] * 40 + [0], dtype=np.uint8))
    out.append(np.array(([0] * 9 + [1]) * 7 + [0] * 5, dtype=np.uint8))
    out.append(rng.integers(0, 3, 64).astype(np.uint8))
    out.append(rng.integers(0, 3, 65).astype(np.uint8))
    out.append(rng.integers(0, 2, 129).astype(np.uint8))
    return out


def main():
    d = {}
    ss = streams()
    for i, s in enumerate(ss):
        blk = rc.tunstall_compress_block(s)
        d["input_%02d" % i] = s
        d["block_%02d" % i] = blk
        ns = int(blk[0])
        print("%2d: %6d symbols, %3d distinct -> block %6d B" % (i, len(s), ns, len(blk)))
    d["count"] = np.array(len(ss))
    np.savez_compressed(os.path.join(OUT, "tunstall_enc_kat.npz"), **d)


if __name__ == "__main__":
    main()
