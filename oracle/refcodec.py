"""ctypes binding of oracle/_ref/libcorto_ref.so (the UNMODIFIED reference compiled by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (corto_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libcorto_ref.so")

FLOAT, INT16 = 6, 3          # VertexAttribute::Format (include/corto/vertex_attribute.h:32)
DIFF, ESTIMATED, BORDER = 0, 1, 2   # NormalAttr::Prediction (include/corto/normal_attribute.h:40-42)


class _Mesh(C.Structure):
    _fields_ = [
        ("nvert", C.c_uint32), ("nface", C.c_uint32),
        ("position", C.c_void_p), ("index", C.c_void_p),
        ("position_bits", C.c_int32), ("position_q", C.c_float),
        ("normal", C.c_void_p), ("normal_bits", C.c_int32), ("normal_prediction", C.c_int32),
        ("color", C.c_void_p), ("color_components", C.c_int32), ("color_bits", C.c_int32 * 4),
        ("uv", C.c_void_p), ("uv_q", C.c_float),
        ("radius", C.c_void_p), ("radius_q", C.c_float),
        ("group_end", C.c_void_p), ("ngroups", C.c_uint32),
        ("entropy", C.c_int32),
        ("exif", C.c_char_p), ("nexif", C.c_uint32),
        ("group_nprops", C.c_void_p), ("group_props", C.c_char_p),
    ]


class _Out(C.Structure):
    _fields_ = [
        ("position", C.c_void_p), ("normal", C.c_void_p), ("normal_format", C.c_int32),
        ("color", C.c_void_p), ("color_components", C.c_int32),
        ("uv", C.c_void_p), ("radius", C.c_void_p),
        ("index32", C.c_void_p), ("index16", C.c_void_p),
    ]


_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_last_error.restype = C.c_char_p
        _lib.ref_encode.restype = C.c_int64
        _lib.ref_tunstall_compress_block.restype = C.c_int64
        _lib.ref_groups.restype = C.c_int64
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def encode(mesh, position_bits=14, position_q=0.0, normal_bits=10, normal_prediction=BORDER,
           color_bits=(6, 7, 6, 5), uv_bits=12, radius_q=1.0, entropy=1, exif=None,
           with_normal=True, with_color=True, with_uv=True) -> np.ndarray:
    """Reference crt::Encoder -> .crt bytes (uint8 array, 4-byte aligned base)."""
    m = _Mesh()
    m.nvert, m.nface = mesh.nvert, mesh.nface
    keep = [mesh.position]
    m.position = _ptr(mesh.position)
    m.index = _ptr(mesh.index)
    m.position_bits, m.position_q = position_bits, position_q
    if with_normal and mesh.normal is not None:
        m.normal = _ptr(mesh.normal); m.normal_bits = normal_bits; m.normal_prediction = normal_prediction
    if with_color and mesh.color is not None:
        m.color = _ptr(mesh.color); m.color_components = mesh.color.shape[1]
        for k in range(4):
            m.color_bits[k] = color_bits[k]
    if with_uv and mesh.uv is not None:
        m.uv = _ptr(mesh.uv); m.uv_q = float(np.float32(2.0) ** np.float32(-uv_bits))
    if mesh.radius is not None:
        m.radius = _ptr(mesh.radius); m.radius_q = radius_q
    if mesh.groups is not None:
        g = np.ascontiguousarray(mesh.groups, dtype=np.uint32); keep.append(g)
        m.group_end = _ptr(g); m.ngroups = len(g)
        props = getattr(mesh, "group_props", None)
        if props:
            cnt = np.array([len(d) for d in props], dtype=np.uint32); keep.append(cnt)
            gflat = b"".join(k.encode() + b"\0" + v.encode() + b"\0" for d in props for k, v in d.items())
            m.group_nprops = _ptr(cnt); m.group_props = gflat
    m.entropy = entropy
    if exif:
        flat = b"".join(k.encode() + b"\0" + v.encode() + b"\0" for k, v in exif.items())
        m.exif = flat; m.nexif = len(exif)
    cap = 64 + mesh.nvert * 64 + mesh.nface * 16 + 4096
    out = np.zeros(cap, dtype=np.uint8)
    nv, nf = C.c_uint32(), C.c_uint32()
    n = lib().ref_encode(C.byref(m), _ptr(out), C.c_int64(cap), C.byref(nv), C.byref(nf))
    if n < 0:
        raise RuntimeError("ref_encode: " + lib().ref_last_error().decode())
    if n > cap:
        out = np.zeros(n, dtype=np.uint8)
        lib().ref_encode(C.byref(m), _ptr(out), C.c_int64(n), C.byref(nv), C.byref(nf))
    return aligned_copy(out[:n])


def aligned_copy(b: np.ndarray, align=16) -> np.ndarray:
    """uint8 copy whose base address is `align`-byte aligned (Decoder needs 4, src/decoder.cpp:43)."""
    raw = np.zeros(len(b) + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    v = raw[off:off + len(b)]
    v[:] = b
    return v


def groups(blob: np.ndarray):
    """what the reference Decoder leaves in index.groups after decode(): [(end, {key: value}), ...]"""
    cap = 1 << 16
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().ref_groups(_ptr(blob), len(blob), _ptr(out), C.c_int64(cap))
    if n < 0:
        raise RuntimeError("ref_groups: " + lib().ref_last_error().decode())
    raw = out[:n].tobytes()
    ng = int.from_bytes(raw[:4], "little"); p = 4
    res = []
    for _ in range(ng):
        end = int.from_bytes(raw[p:p + 4], "little"); cnt = int.from_bytes(raw[p + 4:p + 8], "little"); p += 8
        d = {}
        for _ in range(cnt):
            e = raw.index(b"\0", p); k = raw[p:e].decode(); p = e + 1
            e = raw.index(b"\0", p); v = raw[p:e].decode(); p = e + 1
            d[k] = v
        res.append((end, d))
    return res


def probe(blob: np.ndarray):
    nv, nf, mask, cc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32(0)
    r = lib().ref_probe(_ptr(blob), len(blob), C.byref(nv), C.byref(nf), C.byref(mask), C.byref(cc))
    if r != 0:
        raise RuntimeError("ref_probe: " + lib().ref_last_error().decode())
    return dict(nvert=nv.value, nface=nf.value, position=bool(mask.value & 1), normal=bool(mask.value & 2),
                color=bool(mask.value & 4), uv=bool(mask.value & 8), radius=bool(mask.value & 16),
                color_components=cc.value)


def _alloc_outputs(info, normal_format=FLOAT, color_components=4, index16=False, fill=0):
    nv, nf = info["nvert"], info["nface"]
    outs = {}
    o = _Out()
    if info["position"]:
        outs["position"] = np.full((nv, 3), fill, dtype=np.float32); o.position = _ptr(outs["position"])
    if info["normal"]:
        outs["normal"] = np.full((nv, 3), fill, dtype=np.float32 if normal_format == FLOAT else np.int16)
        o.normal = _ptr(outs["normal"]); o.normal_format = normal_format
    if info["color"]:
        # decode happens in place with N input components, so the buffer must hold max(N, out) per vertex
        width = max(color_components, info["color_components"])
        outs["color"] = np.full((nv, width), fill, dtype=np.uint8)
        o.color = _ptr(outs["color"]); o.color_components = color_components
    if info["uv"]:
        outs["uv"] = np.full((nv, 2), fill, dtype=np.float32); o.uv = _ptr(outs["uv"])
    if info["radius"]:
        outs["radius"] = np.full((nv, 1), fill, dtype=np.float32); o.radius = _ptr(outs["radius"])
    if nf:
        if index16:
            outs["index"] = np.full((nf, 3), fill, dtype=np.uint16); o.index16 = _ptr(outs["index"])
        else:
            outs["index"] = np.full((nf, 3), fill, dtype=np.uint32); o.index32 = _ptr(outs["index"])
    return outs, o


def decode(blob: np.ndarray, normal_format=FLOAT, color_components=4, index16=False):
    """Reference crt::Decoder -> dict of numpy arrays."""
    info = probe(blob)
    outs, o = _alloc_outputs(info, normal_format, color_components, index16)
    r = lib().ref_decode(_ptr(blob), len(blob), C.byref(o))
    if r != 0:
        raise RuntimeError("ref_decode: " + lib().ref_last_error().decode())
    if "color" in outs and outs["color"].shape[1] != color_components:
        # in-place 4->3 / 3->4: the meaningful bytes are the first nvert*out_components of the flat buffer
        outs["color"] = outs["color"].reshape(-1)[: info["nvert"] * color_components].reshape(-1, color_components).copy()
    outs["nvert"], outs["nface"] = info["nvert"], info["nface"]
    return outs


def decode_attr_format(blob: np.ndarray, name: str, fmt: int, n_components: int, fill=0xCD) -> np.ndarray:
    """The reference Decoder with ONE generic attribute bound through setAttribute(name, buffer, format): the nvert*N*8 bytes of the buffer
    afterwards (the decode works in place on int32 values; only DOUBLE uses the second half), prefilled with `fill`."""
    info = probe(blob)
    buf = np.full(info["nvert"] * n_components * 8, fill, dtype=np.uint8)
    idx = np.zeros((max(info["nface"], 1), 3), dtype=np.uint32)
    r = lib().ref_decode_attr_format(_ptr(blob), len(blob), name.encode(), int(fmt), _ptr(buf), _ptr(idx) if info["nface"] else None)
    if r != 0:
        raise RuntimeError("ref_decode_attr_format: " + lib().ref_last_error().decode())
    return buf


def decode_trace(blob: np.ndarray, normal_format=FLOAT, color_components=4, index16=False):
    """decode() + the reference's own topology intermediates (CLERS symbols, prediction triples)."""
    info = probe(blob)
    outs, o = _alloc_outputs(info, normal_format, color_components, index16)
    cap = 4 * info["nface"] + 64
    clers = np.zeros(cap, dtype=np.uint8)
    pred = np.zeros((info["nvert"], 3), dtype=np.uint32)
    n, mf = C.c_uint32(), C.c_uint32()
    r = lib().ref_decode_trace(_ptr(blob), len(blob), C.byref(o), _ptr(clers), cap, C.byref(n), _ptr(pred), C.byref(mf))
    if r != 0:
        raise RuntimeError("ref_decode_trace: " + lib().ref_last_error().decode())
    outs["_clers"] = clers[: n.value].copy()
    outs["_prediction"] = pred
    outs["_max_front"] = mf.value
    outs["nvert"], outs["nface"] = info["nvert"], info["nface"]
    return outs


def decode_timed(blob: np.ndarray, iters=20, normal_format=FLOAT, color_components=4):
    info = probe(blob)
    outs, o = _alloc_outputs(info, normal_format, color_components)
    ns = np.zeros(iters, dtype=np.int64)
    r = lib().ref_decode_timed(_ptr(blob), len(blob), C.byref(o), iters, _ptr(ns))
    if r != 0:
        raise RuntimeError("ref_decode_timed: " + lib().ref_last_error().decode())
    return ns, info


def decode_mt(blobs, nthreads: int, seconds: float, normal_format=FLOAT, color_components=4) -> int:
    """all blobs must have the attribute set of blobs[0]; returns completed decodes over `nthreads` C++ threads"""
    infos = [probe(b) for b in blobs]
    outs, o = _alloc_outputs(infos[0], normal_format, color_components)
    ptrs = (C.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
    lens = (C.c_int * len(blobs))(*[len(b) for b in blobs])
    f = lib().ref_decode_mt
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_double]
    return int(f(ptrs, lens, len(blobs), C.byref(o), max(i["nvert"] for i in infos), max(i["nface"] for i in infos), nthreads, seconds))


def tunstall_tables(probs: np.ndarray):
    probs = np.ascontiguousarray(probs, dtype=np.uint8).reshape(-1, 2)
    idx = np.zeros(256, dtype=np.int32); ln = np.zeros(256, dtype=np.int32)
    tab = np.zeros(8192, dtype=np.uint8); ts = C.c_int32()
    lib().ref_tunstall_tables(_ptr(probs), len(probs), _ptr(idx), _ptr(ln), _ptr(tab), C.byref(ts))
    return idx, ln, tab[: ts.value].copy()


def tunstall_compress_block(symbols: np.ndarray) -> np.ndarray:
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    cap = 2 * len(symbols) + 1024
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().ref_tunstall_compress_block(_ptr(symbols), len(symbols), _ptr(out), C.c_int64(cap))
    if n < 0:
        raise RuntimeError("ref_tunstall_compress_block overflow")
    return out[:n].copy()


def tunstall_decompress(probs: np.ndarray, data: np.ndarray, size: int) -> np.ndarray:
    probs = np.ascontiguousarray(probs, dtype=np.uint8).reshape(-1, 2)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(size, dtype=np.uint8)
    lib().ref_tunstall_decompress(_ptr(probs), len(probs), _ptr(data), len(data), _ptr(out), size)
    return out
