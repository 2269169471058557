"""ctypes binding of oracle/libcorto_oracle.so (oracle/corto_oracle.c, our plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcorto_oracle.so")

FMT_UINT32, FMT_INT32, FMT_UINT16, FMT_INT16, FMT_UINT8, FMT_INT8, FMT_FLOAT, FMT_DOUBLE = range(8)
MAX_ATTRS, NAME_MAX = 16, 64


class AttrInfo(C.Structure):
    _fields_ = [("name", C.c_char * NAME_MAX), ("codec", C.c_uint32), ("q", C.c_float),
                ("N", C.c_uint32), ("format", C.c_uint32), ("strategy", C.c_uint32)]


class Header(C.Structure):
    _fields_ = [("version", C.c_uint32), ("entropy", C.c_uint32), ("nexif", C.c_uint32), ("nattr", C.c_uint32),
                ("nvert", C.c_uint32), ("nface", C.c_uint32), ("body_offset", C.c_uint32),
                ("attr", AttrInfo * MAX_ATTRS)]


class Tunstall(C.Structure):
    _fields_ = [("n", C.c_int), ("sym", C.c_uint8 * 256), ("prob", C.c_uint8 * 256),
                ("index", C.c_uint32 * 256), ("length", C.c_uint32 * 256),
                ("table", C.c_uint8 * (8192 + 512)), ("table_size", C.c_uint32)]


class Binding(C.Structure):
    _fields_ = [("name", C.c_char_p), ("buffer", C.c_void_p), ("format", C.c_uint32), ("out_components", C.c_uint32)]


class Outputs(C.Structure):
    _fields_ = [("bind", C.POINTER(Binding)), ("nbind", C.c_uint32), ("index32", C.c_void_p), ("index16", C.c_void_p)]


class Trace(C.Structure):
    _fields_ = [("clers", C.c_void_p), ("nclers", C.c_uint32), ("nclers_cap", C.c_uint32),
                ("prediction", C.c_void_p), ("max_front", C.c_uint32), ("front_size", C.c_uint32),
                ("attr_raw", C.c_void_p * MAX_ATTRS), ("attr_delta", C.c_void_p * MAX_ATTRS),
                ("normal_ndiffs", C.c_uint32), ("tunstall_in", C.c_uint64), ("tunstall_out", C.c_uint64),
                ("nstreams", C.c_uint32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.co_strerror.restype = C.c_char_p
        _lib.co_bits.restype = C.c_uint32
        _lib.co_bits.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def parse_header(blob: np.ndarray) -> dict:
    h = Header()
    r = lib().co_parse_header(_ptr(blob), C.c_size_t(len(blob)), C.byref(h))
    if r != 0:
        raise RuntimeError(lib().co_strerror(r).decode())
    attrs = [dict(name=h.attr[i].name.decode(), codec=h.attr[i].codec, q=h.attr[i].q, N=h.attr[i].N,
                  format=h.attr[i].format, strategy=h.attr[i].strategy) for i in range(h.nattr)]
    return dict(version=h.version, entropy=h.entropy, nexif=h.nexif, nvert=h.nvert, nface=h.nface,
                body_offset=h.body_offset, attrs=attrs)


def tunstall_tables(probs: np.ndarray):
    probs = np.ascontiguousarray(probs, dtype=np.uint8).reshape(-1, 2)
    t = Tunstall()
    lib().co_tunstall_build(C.byref(t), _ptr(probs), len(probs))
    return (np.array(t.index[:], dtype=np.int32), np.array(t.length[:], dtype=np.int32),
            np.frombuffer(bytes(t.table), dtype=np.uint8)[: t.table_size].copy())


def tunstall_decompress(probs: np.ndarray, data: np.ndarray, size: int) -> np.ndarray:
    probs = np.ascontiguousarray(probs, dtype=np.uint8).reshape(-1, 2)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    t = Tunstall()
    lib().co_tunstall_build(C.byref(t), _ptr(probs), len(probs))
    out = np.zeros(size, dtype=np.uint8)
    lib().co_tunstall_decode(C.byref(t), _ptr(data), len(data), _ptr(out), size)
    return out


def bits(words: np.ndarray, bitoff: int, n: int) -> int:
    return lib().co_bits(_ptr(words), bitoff, n)


def decode_attr_format(blob: np.ndarray, name: str, fmt: int, fill=0xCD) -> np.ndarray:
    """ONE generic attribute bound with output format `fmt` (crt::Decoder::setAttribute(name, buffer, format)): the nvert*N*8 bytes of its
    buffer after the decode, prefilled with `fill` - what refcodec.decode_attr_format returns for the reference."""
    h = parse_header(blob)
    a = [x for x in h["attrs"] if x["name"] == name][0]
    buf = np.full(h["nvert"] * a["N"] * 8, fill, dtype=np.uint8)
    b = (Binding * 1)(Binding(name.encode(), _ptr(buf), fmt, 0))
    o = Outputs(b, 1, None, None)
    idx = np.zeros((max(h["nface"], 1), 3), dtype=np.uint32)
    if h["nface"]:
        o.index32 = _ptr(idx)
    r = lib().co_decode(_ptr(blob), C.c_size_t(len(blob)), C.byref(o), None)
    if r != 0:
        raise RuntimeError(lib().co_strerror(r).decode())
    return buf


def decode(blob: np.ndarray, normal_format=FMT_FLOAT, color_components=4, index16=False, trace=False,
           bind=None, fill=0):
    """Whole decode by the C restatement -> dict of numpy arrays (same keys as refcodec.decode).
    bind: optional set of attribute names to bind (default: all)."""
    h = parse_header(blob)
    nv, nf = h["nvert"], h["nface"]
    outs, binds, keep = {}, [], []
    for a in h["attrs"]:
        name = a["name"]
        if bind is not None and name not in bind:
            continue
        codec = a["codec"]
        if codec == 2:
            arr = np.full((nv, 3), fill, dtype=np.float32 if normal_format == FMT_FLOAT else np.int16)
            b = Binding(name.encode(), _ptr(arr), normal_format, 0)
        elif codec == 3:
            arr = np.full((nv, max(color_components, a["N"])), fill, dtype=np.uint8)
            b = Binding(name.encode(), _ptr(arr), FMT_UINT8, color_components)
        else:
            arr = np.full((nv, a["N"]), fill, dtype=np.float32)
            b = Binding(name.encode(), _ptr(arr), FMT_FLOAT, 0)
        outs[name] = arr
        binds.append(b)
    barr = (Binding * max(len(binds), 1))(*binds)
    o = Outputs(barr, len(binds), None, None)
    if nf:
        outs["index"] = np.full((nf, 3), fill, dtype=np.uint16 if index16 else np.uint32)
        if index16:
            o.index16 = _ptr(outs["index"])
        else:
            o.index32 = _ptr(outs["index"])
    tr = None
    if trace:
        tr = Trace()
        cap = 4 * nf + 64
        outs["_clers"] = np.zeros(cap, dtype=np.uint8)
        outs["_prediction"] = np.zeros((nv, 3), dtype=np.uint32)
        tr.clers = _ptr(outs["_clers"]); tr.nclers_cap = cap
        tr.prediction = _ptr(outs["_prediction"])
        for i, a in enumerate(h["attrs"]):
            n = 2 if a["codec"] == 2 else a["N"]
            outs["_raw_" + a["name"]] = np.zeros((nv, n), dtype=np.int32)
            outs["_delta_" + a["name"]] = np.zeros((nv, n), dtype=np.int32)
            tr.attr_raw[i] = _ptr(outs["_raw_" + a["name"]])
            tr.attr_delta[i] = _ptr(outs["_delta_" + a["name"]])
    r = lib().co_decode(_ptr(blob), C.c_size_t(len(blob)), C.byref(o), C.byref(tr) if tr is not None else None)
    if r != 0:
        raise RuntimeError(lib().co_strerror(r).decode())
    for a in h["attrs"]:
        if a["codec"] == 3 and a["name"] in outs and outs[a["name"]].shape[1] != color_components:
            outs[a["name"]] = outs[a["name"]].reshape(-1)[: nv * color_components].reshape(-1, color_components).copy()
    if trace:
        outs["_clers"] = outs["_clers"][: tr.nclers]
        outs["_trace"] = dict(nclers=tr.nclers, max_front=tr.max_front, front_size=tr.front_size,
                              normal_ndiffs=tr.normal_ndiffs, tunstall_in=tr.tunstall_in,
                              tunstall_out=tr.tunstall_out, nstreams=tr.nstreams)
    outs["nvert"], outs["nface"] = nv, nf
    return outs
