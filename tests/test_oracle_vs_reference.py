"""CPU, build container only: the C restatement against the UNMODIFIED reference (oracle/_ref) on a
randomised corpus much larger than the committed fixtures.  Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from corto_amd import synth
from oracle import oracle as oc
from oracle import refcodec as rc

pytestmark = pytest.mark.skipif(not rc.available(), reason="oracle/_ref not built (needs /root/reference)")


def same(r, o):
    return [k for k in r if isinstance(r[k], np.ndarray) and not k.startswith("_")
            and (r[k].dtype != o[k].dtype or r[k].tobytes() != o[k].tobytes())]


def corpus():
    S = synth
    out = []
    for seed in range(6):
        out.append(("sphere%d" % seed, S.bumpy_sphere(8 + 7 * seed, 4 + 5 * seed, seed), {}))
        out.append(("disc%d" % seed, S.shuffled(S.holey_disc(6 + 6 * seed, seed, hole_frac=0.05 + 0.05 * seed), seed), {}))
        out.append(("torus%d" % seed, S.torus(6 + 5 * seed, 4 + 3 * seed, seed), {}))
        out.append(("closed%d" % seed, S.closed_sphere(5 + 4 * seed, 3 + 3 * seed, seed), {}))
        # non-lattice connectivity (round 5)
        out.append(("ico%d" % seed, S.icosphere(seed % 4, seed), {}))
        out.append(("delaunay%d" % seed, S.delaunay_disc(60 + 350 * seed, seed, holes=2 * seed), {}))
        out.append(("cone%d" % seed, S.cone_fan(7 + 40 * seed, 1 + seed % 4, seed, closed=bool(seed & 1)), {}))
        out.append(("decimated%d" % seed, S.decimated(S.icosphere(1 + seed % 3, seed), keep=0.3 + 0.1 * seed, seed=seed), {}))
        out.append(("confetti%d" % seed, S.shuffled(S.confetti(20 + 60 * seed, seed), seed), {}))
        # non-manifold input (round 6): fins, duplicated / reversed faces, bow-tie vertices, glued pairs (encoder.cpp:450-504,633-636).  Back-to-back pairs
        # put a 0/0 into the estimated normal of their own vertex: the reference's bytes there (x86 cvttss2si of a NaN) are what the oracle must give too
        out.append(("nonmanifold%d" % seed, S.non_manifold([S.delaunay_disc(80 + 200 * seed, seed, holes=seed), S.bumpy_sphere_flipped(8 + 5 * seed, 5 + 2 * seed, seed), S.icosphere(1 + seed % 3, seed)][seed % 3],
                                                           seed=seed, fins=3 + 4 * seed, dups=2 * seed, reversed_dups=1 + 3 * seed, bowties=seed, glue=seed % 4, shuffle_faces=bool(seed & 1)), {}))
    out.append(("fields31", S.full_width_values(S.bumpy_sphere(12, 9, 4), 4, magnitude=2.0 ** 28.6), dict(position_q=1.0, uv_bits=0)))
    out.append(("fields32", S.full_width_values(S.bumpy_sphere(12, 9, 3), 3), dict(position_q=1.0, uv_bits=0)))
    out.append(("merge", S.merge([S.closed_sphere(9, 5, 1), S.closed_sphere(7, 4, 2), S.torus(8, 5, 3), S.holey_disc(9, 4, color_components=4)]), {}))
    return out


@pytest.mark.parametrize("pred", [rc.DIFF, rc.ESTIMATED, rc.BORDER])
def test_meshes_all_normal_modes(pred):
    for name, m, kw in corpus():
        cc = m.color.shape[1]
        for bits in (10, 14, 18):
            blob = rc.encode(m, position_bits=0 if "position_q" in kw else bits, normal_prediction=pred, **kw)
            r = rc.decode_trace(blob, color_components=cc)
            o = oc.decode(blob, color_components=cc, trace=True)
            assert same(r, o) == [], (name, bits)
            assert np.array_equal(r["_clers"], o["_clers"])
            r16 = rc.decode(blob, normal_format=rc.INT16, color_components=cc, index16=True)
            o16 = oc.decode(blob, normal_format=oc.FMT_INT16, color_components=cc, index16=True)
            assert same(r16, o16) == [], (name, bits, "i16")


def test_point_clouds():
    for seed, (nu, nv) in enumerate([(3, 2), (17, 9), (64, 40), (200, 90)]):
        m = synth.point_cloud(nu, nv, seed)
        for pred in (rc.DIFF, rc.BORDER):
            blob = rc.encode(m, normal_prediction=pred)
            assert same(rc.decode(blob), oc.decode(blob)) == []


def test_rgb3_to_rgba4_and_unbound_attributes():
    m = synth.bumpy_sphere(20, 10, 5, color_components=3)
    blob = rc.encode(m, normal_prediction=rc.ESTIMATED)
    r = rc.decode(blob, color_components=4); o = oc.decode(blob, color_components=4)
    assert same(r, o) == [] and (o["color"][:, 3] == 248).all()      # (uchar)(255*8), SURVEY a14
    # only position + index bound: other streams must still be walked correctly
    o2 = oc.decode(blob, bind={"position"})
    assert o2["position"].tobytes() == r["position"].tobytes() and o2["index"].tobytes() == r["index"].tobytes()


def test_tunstall_tables_random():
    rng = np.random.default_rng(7)
    for t in range(1500):
        n = int(rng.integers(2, 256)) if t % 4 == 0 else int(rng.integers(2, 30))
        kind = t % 5
        if kind == 0:
            p = np.sort(rng.integers(0, 256, n))[::-1]
        elif kind == 1:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 0.3)).astype(int))[::-1]
        elif kind == 2:
            p = np.array([max(254 - n, 1)] + [1] * (n - 1))
        elif kind == 3:
            p = np.sort((255 * rng.dirichlet(np.ones(n) * 5)).astype(int))[::-1]
        else:
            p = np.array([250] + list(np.sort(rng.integers(0, 5, n - 1))[::-1]))
        probs = np.stack([rng.permutation(256)[:n], np.clip(p, 0, 255)], 1).astype(np.uint8)
        a = rc.tunstall_tables(probs); b = oc.tunstall_tables(probs)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (t, n, p[:6])


def test_tunstall_streams_random():
    rng = np.random.default_rng(11)
    for t in range(200):
        n = int(rng.integers(1, 6000))
        k = int(rng.integers(1, 20))
        if t % 3 == 0:
            sym = (rng.random(n) < 0.01 * (t % 7)).astype(np.uint8) * int(rng.integers(1, 200))
        else:
            sym = np.minimum(rng.geometric(0.2 + 0.6 * rng.random(), n), k).astype(np.uint8)
        blk = rc.tunstall_compress_block(sym)
        ns = int(blk[0])
        size = int.from_bytes(blk[1 + 2 * ns:5 + 2 * ns].tobytes(), "little")
        cs = int.from_bytes(blk[5 + 2 * ns:9 + 2 * ns].tobytes(), "little")
        out = oc.tunstall_decompress(blk[1:1 + 2 * ns], blk[9 + 2 * ns:9 + 2 * ns + cs], size)
        assert np.array_equal(out, sym), t
