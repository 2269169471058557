"""CPU: the C restatement (oracle/corto_oracle.c) against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py).  This is what pins the oracle on hosts where /root/reference is absent."""
import hashlib

import numpy as np
import pytest

import os

from conftest import ALL_CASES, GOLDEN, aligned, load_golden
from oracle import oracle as oc


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_matches_reference_outputs(name):
    g = load_golden(name)
    cc = int(g["color_components"])
    o = oc.decode(g["crt"], color_components=cc, trace=True)
    for k in ("position", "normal", "color", "uv", "radius", "index"):
        if k in g:
            assert o[k].dtype == g[k].dtype and o[k].shape == g[k].shape, k
            assert o[k].tobytes() == g[k].tobytes(), k
    if "index" in g:
        assert np.array_equal(o["_clers"], g["_clers"])
        assert np.array_equal(o["_prediction"][1:], g["_prediction"][1:])
        assert o["_trace"]["max_front"] == int(g["_max_front"])
        assert o["_trace"]["front_size"] <= int(g["_max_front"])


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_int16_normals_and_u16_index(name):
    g = load_golden(name)
    cc = int(g["color_components"])
    o = oc.decode(g["crt"], normal_format=oc.FMT_INT16, color_components=cc, index16=True)
    if "normal_i16" in g:
        assert o["normal"].tobytes() == g["normal_i16"].tobytes()
    if "index_u16_sha256" in g:
        assert sha(o["index"]) == g["index_u16_sha256"].tobytes().decode()


def test_oracle_mid_mesh_digests():
    g = load_golden("mid34k_digest")
    o = oc.decode(g["crt"])
    assert o["nvert"] == int(g["nvert"]) and o["nface"] == int(g["nface"])
    for k in ("position", "normal", "color", "uv", "index"):
        assert sha(o[k]) == g[k + "_sha256"].tobytes().decode(), k


def test_oracle_c4_blob_digests():
    g = load_golden("c4_blobs16")
    from conftest import aligned
    for seed in range(16):
        o = oc.decode(aligned(g["crt_%02d" % seed]))
        for k in ("position", "normal", "color", "uv", "index"):
            assert sha(o[k]) == g["%s_sha256_%02d" % (k, seed)].tobytes().decode(), (seed, k)


def test_tunstall_tables_kat():
    g = load_golden_kat()
    for i in range(int(g["count"])):
        idx, ln, tab = oc.tunstall_tables(g["probs_%02d" % i])
        assert np.array_equal(idx, g["index_%02d" % i]), i
        assert np.array_equal(ln, g["length_%02d" % i]), i
        assert np.array_equal(tab, g["table_%02d" % i]), i


def test_tunstall_streams_kat():
    g = load_golden_kat()
    for i in range(8):
        blk, sym = g["stream_block_%d" % i], g["stream_symbols_%d" % i]
        ns = int(blk[0])
        size = int.from_bytes(blk[1 + 2 * ns:5 + 2 * ns].tobytes(), "little")
        cs = int.from_bytes(blk[5 + 2 * ns:9 + 2 * ns].tobytes(), "little")
        assert size == len(sym)
        out = oc.tunstall_decompress(blk[1:1 + 2 * ns], blk[9 + 2 * ns:9 + 2 * ns + cs], size)
        assert np.array_equal(out, sym), i


def load_golden_kat():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "tunstall_kat.npz"))
    return {k: z[k] for k in z.files}


def test_bit_fields_random_access_equals_sequential():
    rng = np.random.default_rng(3)
    widths = rng.integers(0, 33, 500)
    vals = [int(rng.integers(0, 1 << w)) if w else 0 for w in widths]
    bits = "".join(format(v, "0%db" % w) if w else "" for v, w in zip(vals, widths))
    bits += "0" * ((-len(bits)) % 32 + 32)
    words = np.array([int(bits[i:i + 32], 2) for i in range(0, len(bits), 32)], dtype=np.uint32)
    off = 0
    for v, w in zip(vals, widths):
        assert oc.bits(words, off, int(w)) == v
        off += int(w)


def test_header_errors():
    g = load_golden("c4_unit")
    bad = g["crt"].copy(); bad[0] ^= 0xFF
    from conftest import aligned
    with pytest.raises(RuntimeError, match="Not a crt file"):
        oc.parse_header(aligned(bad))
    mis = np.zeros(len(g["crt"]) + 17, dtype=np.uint8)
    off = (-mis.ctypes.data) % 16 + 1
    v = mis[off:off + len(g["crt"])]; v[:] = g["crt"]
    with pytest.raises(RuntimeError, match="alignegned"):
        oc.parse_header(v)
    h = oc.parse_header(g["crt"])
    assert [a["name"] for a in h["attrs"]] == ["color", "normal", "position", "uv"]
    assert h["nvert"] == 2112 and h["nface"] == 4096


def test_generic_attribute_output_formats_match_the_reference():
    """GenericAttr::dequantize's integer and DOUBLE branches (include/corto/vertex_attribute.h:195-228; reached through
    Decoder::setAttribute(name, buffer, format)): the C restatement leaves the same bytes in the caller's buffer as the compiled reference
    did (tests/golden/generic_formats.npz, made by make_generic_formats.py), every format, every byte of the nvert*N*8-byte buffer"""
    z = np.load(os.path.join(GOLDEN, "generic_formats.npz"))
    for name in z["cases"].tobytes().decode().split(","):
        blob = aligned(z["crt_" + name])
        for key in [k for k in z.files if k.startswith(name + ".")]:
            _, attr, fmt = key.split(".")
            got = oc.decode_attr_format(blob, attr, int(fmt))
            assert got.tobytes() == z[key].tobytes(), key


def test_nonlattice_c4_sized_blobs_against_the_reference_digests():
    """eight C4-sized blobs with no lattice in them (bench.py's `realistic` Delaunay discs, decimated spheres, an icosphere, a cone of fans): the oracle's
    outputs against SHA-256 digests of what the compiled reference decoded (tests/golden/make_golden.py: nonlattice_blobs)"""
    import hashlib
    z = np.load(os.path.join(GOLDEN, "nonlattice_blobs8.npz"))
    for name in z["names"].tobytes().decode().split(","):
        out = oc.decode(aligned(z["crt_" + name]))
        for k in ("position", "normal", "color", "uv", "index"):
            assert hashlib.sha256(np.ascontiguousarray(out[k]).tobytes()).hexdigest() == z["%s_sha256_%s" % (k, name)].tobytes().decode(), (name, k)
