"""CPU: the C-ABI library loads, exports every symbol include/corto_hip.h declares, and its host-only entry
points (header probe, arena layout, error strings) behave like the reference's constructor.  No kernel runs."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import corto_amd as ca
from conftest import ALL_CASES, ROOT, aligned, load_golden


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(ca.LIB_PATH):
        from corto_amd import build
        build.build()
    return ca.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "corto_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(crthip_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(L, n), n
    assert L.crthip_abi_version() == 6


def test_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: include/corto_hip.h and include/corto/emcorto.h compile as strict C99 (no C++, no torch types)"""
    import subprocess
    src = tmp_path / "cabi.c"
    src.write_text('#include "corto_hip.h"\n#include "corto/emcorto.h"\n'
                   'int main(void) { crthip_batch_stats s; crthip_enc_stream e; crthip_blob_info i; (void)s; (void)e; (void)i; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


@pytest.mark.parametrize("name", ALL_CASES)
def test_probe_matches_golden(L, name):
    g = load_golden(name)
    info = ca.probe(g["crt"])
    nv = g["position"].shape[0]
    assert info.nvert == nv
    assert info.nface == (g["index"].shape[0] if "index" in g else 0)
    names = [a["name"] for a in info.attrs()]
    assert names == sorted(names)
    for k in ("position", "normal", "color", "uv", "radius"):
        assert (k in names) == (k in g)


def test_probe_errors_use_reference_messages(L):
    g = load_golden("c4_unit")
    bad = g["crt"].copy(); bad[1] ^= 0x55
    with pytest.raises(ca.CortoError, match="Not a crt file."):
        ca.probe(aligned(bad))
    raw = np.zeros(len(g["crt"]) + 32, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16 + 2
    mis = raw[off:off + len(g["crt"])]; mis[:] = g["crt"]
    with pytest.raises(ca.CortoError, match="alignegned"):
        ca.probe(mis)
    with pytest.raises(ca.CortoError):
        ca.probe(aligned(g["crt"][:40]))            # truncated header


def test_exif_and_groups(L):
    g = load_golden("radius_attr")
    assert ca.probe_exif(g["crt"]) == {"mtllib": "a.mtl", "note": "x"}
    g2 = load_golden("two_groups")
    assert ca.probe_groups(g2["crt"]) == [400, 1024]
    assert ca.probe_groups(load_golden("cloud_diff")["crt"]) == []


def probe_group_props(blob, g):
    n = L_().crthip_probe_group_props(blob.ctypes.data_as(C.c_void_p), len(blob), g, None, 0)
    assert n >= 0
    buf = C.create_string_buffer(max(int(n), 1))
    L_().crthip_probe_group_props(blob.ctypes.data_as(C.c_void_p), len(blob), g, buf, int(n))
    parts = buf.raw[:n].split(b"\0")[:-1] if n else []
    return {parts[i].decode(): parts[i + 1].decode() for i in range(0, len(parts), 2)}


def L_():
    return ca.lib()


def test_group_properties_match_the_reference(L):
    """Group::properties (include/corto/index_attribute.h:89-99): the blob was written by the reference's addGroup(end, props)
    (include/corto/encoder.h:75) and `groups_ref` is what the reference Decoder reported back for it"""
    g = load_golden("group_props")
    blob = aligned(g["crt"])
    want = []
    for line in g["groups_ref"].tobytes().decode().split("\n"):
        f = line.split("\t")
        want.append((int(f[0]), dict(kv.split("=", 1) for kv in f[1:])))
    assert [e for e, _ in want] == [100, 101, 400, 576] and want[2][1] == {"a": "first", "m": "", "z": "last"}
    assert ca.probe_groups(blob) == [e for e, _ in want]
    for i, (_, props) in enumerate(want):
        assert probe_group_props(blob, i) == props


def test_truncated_bodies_are_rejected_on_the_host(L):
    g = load_golden("holey_disc")
    blob = g["crt"]
    for cut in (len(blob) - 1, len(blob) // 2, 130):
        n = L.crthip_probe_groups(aligned(blob[:cut]).ctypes.data_as(C.c_void_p), cut, None, 0)
        assert n == -3, cut                          # CRTHIP_E_TRUNCATED: the walk validates every extent


def test_arena_layout():
    offs, total = ca.arena_layout([5, 16, 17, 0, 3])
    assert list(offs) == [0, 16, 32, 64, 64] and total == 80


def test_no_gpu_means_loud_failure(L):
    if L.crthip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ca.CortoError, match="no CPU fallback"):
        ca.Context(0)


def _veneer():
    from corto_amd import build
    if not os.path.exists(build.VENEER):
        build.build()
    V = C.CDLL(build.VENEER)
    V.CreateDecoder.restype = C.c_void_p
    V.CreateDecoder.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    V.DestroyDecoder.argtypes = [C.c_void_p]
    V.DecodeMesh.restype = C.c_int
    V.DecodeMesh.argtypes = [C.c_void_p] * 6
    return V


def test_unity_veneer_exports_and_parses_on_the_host(L):
    """the legacy C ABI of upstream's cortocodec_unity (src/corto_codec.h:41-43): same three symbols; CreateDecoder is the
    host-side header parse (info = (nface, nvert), src/corto_codec.cpp:11-13) and needs no GPU"""
    V = _veneer()
    for name in ("c4_unit", "cloud_border"):
        g = load_golden(name)
        blob = aligned(g["crt"])
        info = np.zeros(2, dtype=np.float32)
        d = V.CreateDecoder(len(blob), blob.ctypes.data, info.ctypes.data)
        assert d, name
        nface = len(g["index"]) if "index" in g else 0
        assert info[0] == nface and info[1] == len(g["position"])
        if nface == 0:                       # point cloud: refused before anything touches the device
            assert V.DecodeMesh(d, None, None, None, None, None) == -1
        V.DestroyDecoder(d)
    junk = aligned(np.zeros(64, dtype=np.uint8))
    assert not V.CreateDecoder(len(junk), junk.ctypes.data, None)      # "Not a crt file." does not cross the C boundary


EM_SYMBOLS = ("newDecoder ngroups groups nvert nface hasAttr hasNormal hasColor hasUv setPositions setNormals32 setNormals16 "
              "setColors setUvs setIndex16 setIndex32 decode deleteDecoder").split()     # upstream's eighteen
EM_EXTRA = ("lastError",)       # this repo's one addition (a failed decode must not look like a successful one)


def em_veneer():
    from corto_amd import build
    if not os.path.exists(build.EMVENEER):
        build.build()
    E = C.CDLL(build.EMVENEER)
    E.newDecoder.restype = C.c_void_p
    E.newDecoder.argtypes = [C.c_int, C.c_void_p]
    for nm in ("ngroups", "nvert", "nface", "lastError"):
        getattr(E, nm).restype = C.c_int
        getattr(E, nm).argtypes = [C.c_void_p]
    for nm in ("hasNormal", "hasColor", "hasUv"):
        getattr(E, nm).restype = C.c_bool
        getattr(E, nm).argtypes = [C.c_void_p]
    E.hasAttr.restype = C.c_bool
    E.hasAttr.argtypes = [C.c_void_p, C.c_char_p]
    for nm in ("groups", "setPositions", "setNormals32", "setNormals16", "setUvs", "setIndex16", "setIndex32"):
        getattr(E, nm).restype = None
        getattr(E, nm).argtypes = [C.c_void_p, C.c_void_p]
    E.setColors.restype = None
    E.setColors.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for nm in ("decode", "deleteDecoder"):
        getattr(E, nm).restype = None
        getattr(E, nm).argtypes = [C.c_void_p]
    return E


def test_js_veneer_exports_and_parses_on_the_host(L):
    """the flat C ABI of upstream's wasm module (html/js/emscripten/emcorto.cpp:14-89): all eighteen symbols of
    include/corto/emcorto.h; the header queries (nvert, nface, groups, has*) are host-side and need no GPU"""
    hdr = open(os.path.join(ROOT, "include", "corto", "emcorto.h")).read()
    E = em_veneer()
    for nm in EM_SYMBOLS + list(EM_EXTRA):
        assert re.search(r"\b%s\(" % nm, hdr), nm
        assert hasattr(E, nm), nm
    for name in ("two_groups", "cloud_border", "nrm_estimated_rgb"):
        g = load_golden(name)
        blob = aligned(g["crt"])
        d = E.newDecoder(len(blob), blob.ctypes.data)
        assert d, name
        assert E.nvert(d) == len(g["position"]) and E.nface(d) == (len(g["index"]) if "index" in g else 0)
        assert E.hasNormal(d) == ("normal" in g) and E.hasColor(d) == ("color" in g) and E.hasUv(d) == ("uv" in g)
        assert E.hasAttr(d, b"position") and not E.hasAttr(d, b"nope")
        assert E.ngroups(d) == 0              # like upstream, groups are read by decode() (src/decoder.cpp:137,165)
        E.deleteDecoder(d)
    junk = aligned(np.zeros(64, dtype=np.uint8))
    assert not E.newDecoder(len(junk), junk.ctypes.data)
    assert E.lastError(None) == -2            # CRTHIP_E_MAGIC: "Not a crt file." (src/decoder.cpp:51)
    assert not E.newDecoder(len(junk) - 1, junk.ctypes.data + 1)
    assert E.lastError(None) == -1            # CRTHIP_E_ALIGN (src/decoder.cpp:44)
    assert E.nvert(None) == 0 and E.ngroups(None) == 0 and not E.hasUv(None)
    E.decode(None); E.deleteDecoder(None)
