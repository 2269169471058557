#!/usr/bin/env python3
"""Generate the committed golden fixtures FROM THE UNMODIFIED REFERENCE (oracle/_ref/libcorto_ref.so,
built by oracle/Makefile from /root/reference).  Run in the build container only:

    python tests/golden/make_golden.py

Every fixture is DATA: a .crt blob produced by the reference crt::Encoder plus the arrays the reference
crt::Decoder returned for it (or their SHA-256 when large), plus Tunstall known-answer tables from
crt::Tunstall::createDecodingTables2.  The reference has no tests/golden vectors of its own
(SURVEY.md §4), so these outputs of the reference itself are what pins the oracle and the HIP path.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corto_amd import synth            # noqa: E402
from oracle import refcodec as rc      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import cases  # noqa: E402


def tunstall_kat():
    # Tunstall known-answer tables (createDecodingTables2) incl. the low-entropy branch, ties, zero tails
    rng = np.random.default_rng(20250926)
    kats = {}
    plist = [np.array([[65, 254], [66, 0]]), np.array([[0, 255], [1, 0]]), np.array([[3, 128], [9, 127]]),
             np.array([[1, 85], [2, 85], [3, 85]]), np.array([[7, 250], [8, 2], [9, 2], [1, 1]]),
             np.array([[i, 1] for i in range(255)]), np.array([[i, 255 // 40] for i in range(40)])]
    for t in range(57):
        n = int(rng.integers(2, 41))
        kind = t % 4
        if kind == 0:
            p = np.sort(rng.integers(0, 256, n))[::-1]
        elif kind == 1:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 0.3)).astype(int))[::-1]
        elif kind == 2:
            p = np.array([250 - n] + list(np.sort(rng.integers(0, 4, n - 1))[::-1]))
        else:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 4)).astype(int))[::-1]
        plist.append(np.stack([rng.permutation(256)[:n], np.clip(p, 0, 255)], 1))
    # alphabets above 64 symbols (the device's general path: two or three expansions, seeds among the parents)
    rng65 = np.random.default_rng(65)
    for n, kind in ((65, 0), (66, 1), (70, 2), (90, 0), (100, 1), (127, 2), (128, 0), (129, 1), (200, 0)):
        if kind == 0:
            p = np.sort(rng65.integers(1, 8, n))[::-1]
        elif kind == 1:
            p = np.array([120, 60, 30] + [1] * (n - 3))
        else:
            p = np.sort((255 * rng65.dirichlet(np.ones(n) * 0.5)).astype(int))[::-1]
        plist.append(np.stack([rng65.permutation(256)[:n], np.clip(p, 0, 255)], 1))
    for i, pr in enumerate(plist):
        pr = pr.astype(np.uint8)
        idx, ln, tab = rc.tunstall_tables(pr)
        kats["probs_%02d" % i] = pr
        kats["index_%02d" % i] = idx.astype(np.uint16)
        kats["length_%02d" % i] = ln.astype(np.uint16)
        kats["table_%02d" % i] = tab
    kats["count"] = np.array(len(plist))
    # Tunstall stream KATs: symbols -> block (reference compressor) -> symbols (reference decompressor)
    for i in range(8):
        n = [1, 2, 7, 300, 5000, 5000, 20000, 3][i]
        if i == 5:
            sym = (rng.random(n) < 0.004).astype(np.uint8) * 3       # very low entropy -> count>=16 branch
        elif i == 6:
            sym = np.minimum(rng.geometric(0.35, n), 14).astype(np.uint8)
        elif i == 7:
            sym = np.array([5, 5, 5], dtype=np.uint8)                 # single symbol: memset path
        else:
            sym = rng.integers(0, 6, n).astype(np.uint8)
        blk = rc.tunstall_compress_block(sym)
        ns = int(blk[0]); size = int.from_bytes(blk[1 + 2 * ns:5 + 2 * ns].tobytes(), "little")
        cs = int.from_bytes(blk[5 + 2 * ns:9 + 2 * ns].tobytes(), "little")
        back = rc.tunstall_decompress(blk[1:1 + 2 * ns], blk[9 + 2 * ns:9 + 2 * ns + cs], size)
        assert np.array_equal(back, sym), i
        kats["stream_block_%d" % i] = blk
        kats["stream_symbols_%d" % i] = sym
    np.savez_compressed(os.path.join(OUT, "tunstall_kat.npz"), **kats)
    print("tunstall_kat         %d tables, 8 streams" % len(plist))


def main():
    only = set(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else None   # add a case without rewriting the others
    index = []
    for name, mesh, kw in cases():
        if only is not None and name not in only:
            continue
        blob = rc.encode(mesh, **kw)
        cc = mesh.color.shape[1] if mesh.color is not None and kw.get("with_color", True) else 4
        ref = rc.decode_trace(blob, color_components=cc)
        ref16 = rc.decode(blob, normal_format=rc.INT16, color_components=cc, index16=ref["nvert"] < 65536)
        d = {"crt": np.asarray(blob).copy()}
        for k, v in ref.items():
            if isinstance(v, np.ndarray):
                d[k] = v
        d["_max_front"] = np.array(ref["_max_front"])
        if "normal" in ref16:
            d["normal_i16"] = ref16["normal"]
        if "index" in ref16 and ref16["index"].dtype == np.uint16:
            d["index_u16_sha256"] = np.frombuffer(sha(ref16["index"]).encode(), dtype=np.uint8)
        d["color_components"] = np.array(cc)
        # index.groups as the reference Decoder reports them after decode(): "end\tkey=value\tkey=value" per group, one group per line
        d["groups_ref"] = np.frombuffer("\n".join("\t".join([str(e)] + ["%s=%s" % kv for kv in p.items()]) for e, p in rc.groups(blob)).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        index.append((name, len(blob), ref["nvert"], ref["nface"]))
        print("%-20s crt %7d B  nvert %6d nface %6d" % index[-1])

    if only is not None:
        if "delaunay_tris" in only:
            delaunay_store()
        if "tunstall_kat" in only:
            tunstall_kat()
        if "nonlattice_blobs8" in only:
            nonlattice_blobs()
        return
    # mid-size mesh (C1-class, 34 060 verts / 67 600 tris): blob + digests only
    m = synth.bumpy_sphere(260, 130, seed=34)
    blob = rc.encode(m, normal_prediction=rc.BORDER)
    ref = rc.decode(blob)
    d = {"crt": np.asarray(blob).copy()}
    for k, v in ref.items():
        if isinstance(v, np.ndarray):
            d[k + "_sha256"] = np.frombuffer(sha(v).encode(), dtype=np.uint8)
    d["nvert"] = np.array(ref["nvert"]); d["nface"] = np.array(ref["nface"])
    np.savez_compressed(os.path.join(OUT, "mid34k_digest.npz"), **d)
    print("mid34k_digest        crt %7d B  nvert %6d nface %6d" % (len(blob), ref["nvert"], ref["nface"]))

    # 16 distinct C4-unit blobs (seeds 0..15) for bench / batch tests: blobs + per-array digests
    d = {}
    for seed in range(16):
        m = synth.bumpy_sphere(64, 32, seed=seed)
        blob = rc.encode(m, normal_prediction=rc.BORDER)
        ref = rc.decode(blob)
        d["crt_%02d" % seed] = np.asarray(blob).copy()
        for k in ("position", "normal", "color", "uv", "index"):
            d["%s_sha256_%02d" % (k, seed)] = np.frombuffer(sha(ref[k]).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "c4_blobs16.npz"), **d)
    print("c4_blobs16           %d blobs" % 16)

    nonlattice_blobs()
    tunstall_kat()
    delaunay_store()


def delaunay_store():
    """the triangulation of every Delaunay point set a fixture is made from (synth._delaunay_2d looks them up by the points' digest): tests that rebuild
    those meshes then encode what the reference encoded, whatever triangulator is installed"""
    synth._DELAUNAY_STORE.clear()
    synth._DELAUNAY_RECORD = {}
    cases(); nonlattice_meshes()
    np.savez_compressed(os.path.join(OUT, "delaunay_tris.npz"), **synth._DELAUNAY_RECORD)
    print("delaunay_tris        %d point sets" % len(synth._DELAUNAY_RECORD))
    synth._DELAUNAY_RECORD = None


def nonlattice_meshes():
    """C4-sized blobs with no lattice in them - bench.py's `realistic` blobs (Delaunay discs with holes, seeds 0-3) and a decimated sphere, an icosphere, a
    cone of fans: (name, mesh, prediction) - shared with the tests, which rebuild them for the repo's own encoder"""
    out = [("delaunay%d" % sd, synth.delaunay_disc(2310, seed=sd, holes=6 + sd % 5), rc.BORDER) for sd in range(4)]
    out += [("decimated0", synth.decimated(synth.icosphere(4, seed=0), keep=0.8, seed=0), rc.ESTIMATED), ("decimated1", synth.decimated(synth.icosphere(4, seed=1), keep=0.6, seed=1), rc.BORDER),
            ("icosphere4", synth.icosphere(4, seed=2), rc.ESTIMATED), ("cone128", synth.cone_fan(128, 16, seed=3), rc.BORDER)]
    return out


def nonlattice_blobs():
    d = {}
    names = []
    for name, m, pred in nonlattice_meshes():
        blob = rc.encode(m, normal_prediction=pred)
        ref = rc.decode(blob)
        d["crt_" + name] = np.asarray(blob).copy()
        for k in ("position", "normal", "color", "uv", "index"):
            d["%s_sha256_%s" % (k, name)] = np.frombuffer(sha(ref[k]).encode(), dtype=np.uint8)
        names.append(name)
    d["names"] = np.frombuffer(",".join(names).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "nonlattice_blobs8.npz"), **d)
    print("nonlattice_blobs8    %d blobs" % len(names))


if __name__ == "__main__":
    main()
