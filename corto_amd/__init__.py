"""corto_amd — MI355X-native decode path of corto (.crt meshes / point clouds).

Python host mirror of the C ABI in include/corto_hip.h (ctypes over corto_amd/lib/libcorto_hip.so).
PyTorch is used only as plumbing for device memory; the product is the HIP library.

The reference interface this mirrors is crt::Decoder (include/corto/decoder.h:38-73 upstream):
    Decoder(len, input) -> nvert / nface / attributes     here: probe(blob) / Decoder(blob)
    setPositions / setNormals / setColors / setUvs / setAttribute / setIndex
    decode()
plus the batch form that has no upstream equivalent (Batch): many independent blobs, one set of launches.

There is no CPU fallback: if the HIP library or a GPU is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CORTO_HIP_LIB_PATH") or os.path.join(_HERE, "lib", "libcorto_hip.so")   # (the override: A/B probes with a variant build, tools/)

FMT_UINT32, FMT_INT32, FMT_UINT16, FMT_INT16, FMT_UINT8, FMT_INT8, FMT_FLOAT, FMT_DOUBLE = range(8)
CODEC_GENERIC, CODEC_NORMAL, CODEC_COLOR = 1, 2, 3
MAX_ATTRS, NAME_MAX, MAX_KERNELS = 16, 64, 32


class CortoError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class AttrInfo(C.Structure):
    _fields_ = [("name", C.c_char * NAME_MAX), ("codec", C.c_uint32), ("q", C.c_float),
                ("components", C.c_uint32), ("format", C.c_uint32), ("strategy", C.c_uint32)]


class BlobInfo(C.Structure):
    _fields_ = [("version", C.c_uint32), ("entropy", C.c_uint32), ("nvert", C.c_uint32), ("nface", C.c_uint32),
                ("nattr", C.c_uint32), ("attr", AttrInfo * MAX_ATTRS), ("nexif", C.c_uint32), ("body_offset", C.c_uint32)]

    def attrs(self):
        return [dict(name=self.attr[i].name.decode(), codec=self.attr[i].codec, q=self.attr[i].q,
                     components=self.attr[i].components, format=self.attr[i].format, strategy=self.attr[i].strategy)
                for i in range(self.nattr)]


class AttrBinding(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("format", C.c_uint32), ("out_components", C.c_uint32), ("stride", C.c_uint32), ("reserved", C.c_uint32)]


class BatchStats(C.Structure):
    _fields_ = [("arena_bytes", C.c_uint64), ("output_bytes", C.c_uint64), ("tunstall_in", C.c_uint64),
                ("tunstall_out", C.c_uint64), ("tunstall_tables", C.c_uint64), ("tunstall_streams", C.c_uint32),
                ("total_nvert", C.c_uint64), ("total_nface", C.c_uint64), ("scratch_bytes", C.c_uint64),
                ("clers_symbols", C.c_uint64), ("split_bytes", C.c_uint64), ("topology_fallbacks", C.c_uint64),
                ("host_plan_us", C.c_float), ("host_stage_us", C.c_float), ("host_launch_us", C.c_float), ("host_create_us", C.c_float),
                ("topology_scale", C.c_uint32), ("tunstall_dictionaries", C.c_uint32), ("delta_redone", C.c_uint32), ("delta_walked", C.c_uint32), ("delta_wide", C.c_uint32), ("descriptor_bytes", C.c_uint32), ("int16_streams", C.c_uint32)]


class MeshDesc(C.Structure):
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


class PoolItem(C.Structure):
    _fields_ = [("nblobs", C.c_uint32), ("blobs", C.c_void_p), ("lens", C.c_void_p), ("device_arena", C.c_void_p)]


class PoolReport(C.Structure):
    _fields_ = [("elapsed_s", C.c_double), ("steps", C.c_uint64), ("triangles", C.c_uint64), ("vertices", C.c_uint64),
                ("failed_blobs", C.c_uint64), ("first_error", C.c_int32), ("devices_used", C.c_uint32),
                ("steps_per_device", C.c_uint64 * 16), ("topology_fallbacks", C.c_uint64),
                ("poisoned_lanes", C.c_uint32), ("pinned_devices", C.c_uint32), ("host_us_per_step", C.c_float), ("host_plan_max_us", C.c_float),
                ("host_wait_us", C.c_float), ("host_finish_us", C.c_float), ("host_plan_us", C.c_float), ("host_launch_max_us", C.c_float)]


class KernelTimes(C.Structure):
    _fields_ = [("count", C.c_uint32), ("name", C.c_char_p * MAX_KERNELS), ("ms", C.c_float * MAX_KERNELS),
                ("launches", C.c_uint32 * MAX_KERNELS)]

    def as_dict(self):
        return {self.name[i].decode(): dict(ms=float(self.ms[i]), launches=int(self.launches[i])) for i in range(self.count)}


_lib = None


def lib():
    """Load the HIP library.  Fails loudly when it has not been built (python -m corto_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CortoError(-9, "corto_amd: %s is missing - build it with `python -m corto_amd.build` "
                                 "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        # torch (when present) bundles its own libamdhip64.so.7; it must be loaded FIRST so that this library
        # binds to the same HIP runtime - two HIP runtimes in one process cannot both own the GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.crthip_last_error.restype = C.c_char_p
        L.crthip_strerror.restype = C.c_char_p
        L.crthip_abi_version.restype = C.c_uint32
        L.crthip_probe_exif.restype = C.c_int64
        L.crthip_probe_groups.restype = C.c_int64
        L.crthip_probe_group_props.restype = C.c_int64
        L.crthip_probe_group_props.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]
        L.crthip_arena_layout.restype = C.c_uint64
        L.crthip_batch_size.restype = C.c_uint32
        L.crthip_batch_debug_read.restype = C.c_int64
        L.crthip_probe.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(BlobInfo)]
        L.crthip_probe_exif.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.crthip_probe_groups.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.crthip_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.crthip_ctx_destroy.argtypes = [C.c_void_p]
        L.crthip_ctx_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.crthip_ctx_set_single_stream.argtypes = [C.c_void_p, C.c_int]
        L.crthip_ctx_sync.argtypes = [C.c_void_p]
        L.crthip_batch_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.crthip_batch_destroy.argtypes = [C.c_void_p]
        L.crthip_batch_size.argtypes = [C.c_void_p]
        L.crthip_batch_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(BlobInfo)]
        L.crthip_batch_bind.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.crthip_batch_bind_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.crthip_batch_decode.argtypes = [C.c_void_p]
        L.crthip_batch_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.crthip_batch_get_stats.argtypes = [C.c_void_p, C.POINTER(BatchStats)]
        L.crthip_batch_kernel_times.argtypes = [C.c_void_p, C.POINTER(KernelTimes)]
        L.crthip_batch_debug_read.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_size_t]
        L.crthip_decode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32]
        L.crthip_arena_layout.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.crthip_encode.restype = C.c_int64
        L.crthip_encode.argtypes = [C.POINTER(MeshDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.crthip_encode_gpu.restype = C.c_int64
        L.crthip_encode_gpu.argtypes = [C.c_void_p, C.POINTER(MeshDesc), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.crthip_encode_values.restype = C.c_int64
        L.crthip_encode_values.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.crthip_tunstall_encode_blocks.restype = C.c_int64
        L.crthip_tunstall_encode_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.crthip_tunstall_decode_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.POINTER(KernelTimes)]
        L.crthip_pool_create.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.crthip_pool_destroy.argtypes = [C.c_void_p]
        L.crthip_pool_lanes.restype = C.c_uint32
        L.crthip_pool_lanes.argtypes = [C.c_void_p]
        L.crthip_pool_warning.restype = C.c_char_p
        L.crthip_pool_warning.argtypes = [C.c_void_p]
        L.crthip_pool_set_packed_host_blobs.argtypes = [C.c_void_p, C.c_int]
        L.crthip_ctx_set_packed_host_blobs.argtypes = [C.c_void_p, C.c_int]
        L.crthip_pool_set_outputs_to_host.argtypes = [C.c_void_p, C.c_int]
        L.crthip_pool_set_render_layouts.argtypes = [C.c_void_p, C.c_int]
        L.crthip_pool_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(PoolReport), C.c_void_p]
        L.crthip_pool_lane_item.restype = C.c_int64
        L.crthip_pool_lane_item.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.crthip_pool_lane_read.restype = C.c_int64
        L.crthip_pool_lane_read.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _check(code: int):
    if code != 0:
        raise CortoError(code, lib().crthip_last_error().decode())


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def aligned_blob(b, align: int = 16) -> np.ndarray:
    """uint8 copy of b whose base address is `align`-aligned (the decoder needs 4: src/decoder.cpp:43-44)."""
    b = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    raw = np.zeros(len(b) + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    v = raw[off:off + len(b)]
    v[:] = b
    return v


def probe(blob: np.ndarray) -> BlobInfo:
    """Header facts of one blob (host only, no GPU needed)."""
    info = BlobInfo()
    _check(lib().crthip_probe(_np_ptr(blob), len(blob), C.byref(info)))
    return info


def probe_exif(blob: np.ndarray) -> Dict[str, str]:
    n = lib().crthip_probe_exif(_np_ptr(blob), len(blob), None, 0)
    if n < 0:
        _check(int(n))
    buf = C.create_string_buffer(int(n) + 1)
    lib().crthip_probe_exif(_np_ptr(blob), len(blob), buf, n)
    parts = buf.raw[:n].split(b"\0")[:-1]
    return {parts[i].decode(): parts[i + 1].decode() for i in range(0, len(parts), 2)}


def probe_groups(blob: np.ndarray) -> List[int]:
    n = lib().crthip_probe_groups(_np_ptr(blob), len(blob), None, 0)
    if n < 0:
        _check(int(n))
    g = np.zeros(max(int(n), 1), dtype=np.uint32)
    lib().crthip_probe_groups(_np_ptr(blob), len(blob), _np_ptr(g), int(n))
    return [int(x) for x in g[:n]]


def arena_layout(lens: Sequence[int]):
    lens_a = np.asarray(lens, dtype=np.uint32)
    offs = np.zeros(len(lens_a), dtype=np.uint64)
    total = lib().crthip_arena_layout(len(lens_a), _np_ptr(lens_a), _np_ptr(offs))
    return offs, int(total)


DIFF, ESTIMATED, BORDER = 0, 1, 2


def encode(mesh, position_bits=14, position_q=0.0, normal_bits=10, normal_prediction=BORDER, color_bits=(6, 7, 6, 5),
           uv_bits=12, radius_q=1.0, entropy=1, exif=None, with_normal=True, with_color=True, with_uv=True, ctx=None) -> np.ndarray:
    """.crt blob of a corto_amd.synth.Mesh (byte-identical to upstream crt::Encoder, see csrc/encoder.cpp).  Host only by
    default; with ctx=Context the value coding and the entropy coder run on the device (crthip_encode_gpu) - same bytes.
    Same keyword meaning as upstream's CLI: -v position_bits, -n normal_bits, -N prediction, -u uv_bits (src/main.cpp:93-216)."""
    m = MeshDesc()
    m.nvert, m.nface = mesh.nvert, mesh.nface
    keep = []
    m.position = mesh.position.ctypes.data
    if mesh.index is not None:
        m.index = mesh.index.ctypes.data
    m.position_bits, m.position_q = position_bits, position_q
    if with_normal and mesh.normal is not None:
        m.normal = mesh.normal.ctypes.data; m.normal_bits = normal_bits; m.normal_prediction = normal_prediction
    if with_color and mesh.color is not None:
        m.color = mesh.color.ctypes.data; m.color_components = mesh.color.shape[1]
        for k in range(4):
            m.color_bits[k] = color_bits[k]
    if with_uv and mesh.uv is not None:
        m.uv = mesh.uv.ctypes.data; m.uv_q = float(np.float32(2.0) ** np.float32(-uv_bits))
    if mesh.radius is not None:
        m.radius = mesh.radius.ctypes.data; m.radius_q = radius_q
    if mesh.groups is not None:
        g = np.ascontiguousarray(mesh.groups, dtype=np.uint32); keep.append(g)
        m.group_end = g.ctypes.data; m.ngroups = len(g)
        props = getattr(mesh, "group_props", None)          # list of dicts, one per group (Encoder::addGroup(end, props))
        if props:
            cnt = np.array([len(d) for d in props], dtype=np.uint32); keep.append(cnt)
            gflat = b"".join(k.encode() + b"\0" + v.encode() + b"\0" for d in props for k, v in d.items())
            m.group_nprops = cnt.ctypes.data; m.group_props = gflat
    m.entropy = entropy
    if exif:
        flat = b"".join(k.encode() + b"\0" + v.encode() + b"\0" for k, v in exif.items())
        m.exif = flat; m.nexif = len(exif)
    cap = 64 * (mesh.nvert + mesh.nface) + 65536            # one pass unless the estimate is too small
    for _ in range(2):
        out = np.zeros(cap + 16, dtype=np.uint8)
        off = (-out.ctypes.data) % 16
        dst = out[off:].ctypes.data_as(C.c_void_p)
        if ctx is not None:
            n = lib().crthip_encode_gpu(ctx.handle, C.byref(m), dst, cap, None, None)
        else:
            n = lib().crthip_encode(C.byref(m), dst, cap, None, None)
        if n < 0:
            _check(int(n))
        if n <= cap:
            return out[off:off + int(n)]
        cap = int(n)
    raise CortoError(-9, "encode: size changed between calls")


class Context:
    """One per GPU: owns the HIP stream and the scratch pool (crthip_ctx)."""

    def __init__(self, device: int = 0):
        self.handle = C.c_void_p()
        _check(lib().crthip_ctx_create(device, C.byref(self.handle)))
        self.device = device

    def set_profiling(self, on: bool):
        _check(lib().crthip_ctx_set_profiling(self.handle, int(on)))

    def set_single_stream(self, on: bool = True):
        """one HIP stream per context instead of two: faster from about $GPU_MAX_HW_QUEUES / 2 contexts per GPU up (corto_hip.h)"""
        _check(lib().crthip_ctx_set_single_stream(self.handle, int(on)))

    def set_packed_host_blobs(self, on: bool = True):
        """blobs laid out as an arena in ONE pinned host buffer (pinned_host_arena) are uploaded straight from there (corto_hip.h)"""
        _check(lib().crthip_ctx_set_packed_host_blobs(self.handle, int(on)))

    def sync(self):
        _check(lib().crthip_ctx_sync(self.handle))

    def close(self):
        if self.handle:
            lib().crthip_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DT = {"f32": np.float32, "i16": np.int16, "u8": np.uint8, "u32": np.uint32, "u16": np.uint16}


class Batch:
    """A planned batch of independent .crt blobs (crthip_batch)."""

    def __init__(self, ctx: Context, blobs: Sequence[np.ndarray], device_arena=None):
        self.ctx = ctx
        self.handle = C.c_void_p()
        self._plan(blobs, device_arena, create=True)

    def reset(self, blobs: Sequence[np.ndarray], device_arena=None):
        """crthip_batch_reset: re-plan this object for another list of blobs (bindings and outputs are dropped)."""
        self._plan(blobs, device_arena, create=False)

    def _plan(self, blobs, device_arena, create):
        self.blobs = list(blobs)
        n = len(self.blobs)
        self._ptrs = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b in self.blobs])
        self._lens = np.array([len(b) for b in self.blobs], dtype=np.uint32)
        self._arena = device_arena
        arena_ptr = C.c_void_p(device_arena.data_ptr()) if device_arena is not None else None
        if create:
            _check(lib().crthip_batch_create(self.ctx.handle, n, self._ptrs, _np_ptr(self._lens), arena_ptr, C.byref(self.handle)))
        else:
            _check(lib().crthip_batch_reset(self.handle, n, self._ptrs, _np_ptr(self._lens), arena_ptr))
        self.infos = []
        for i in range(n):
            info = BlobInfo()
            _check(lib().crthip_batch_info(self.handle, i, C.byref(info)))
            self.infos.append(info)
        self.outputs: List[Dict[str, object]] = []
        self._keep = None
        self._interleaved = None

    def __len__(self):
        return len(self.blobs)

    # -- outputs -----------------------------------------------------------------------------------
    def allocate_outputs(self, normal_format=FMT_FLOAT, color_components: Optional[int] = None, index16=False,
                         only: Optional[set] = None, fill: Optional[int] = None):
        """Allocate every output of every blob inside ONE device buffer (torch.uint8) and bind them.
        Returns a list (per blob) of dicts name -> torch tensor view."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        plan = []   # (blob, name, offset, nbytes, dtype, shape, binding_slot)
        off = 0

        def take(nbytes):
            nonlocal off
            off = (off + 255) & ~255
            r = off
            off += nbytes
            return r
        for i, info in enumerate(self.infos):
            nv, nf = info.nvert, info.nface
            for k, a in enumerate(info.attrs()):
                if only is not None and a["name"] not in only:
                    continue
                if a["codec"] == CODEC_NORMAL:
                    dt = "f32" if normal_format == FMT_FLOAT else "i16"
                    plan.append((i, a["name"], take(nv * 3 * np.dtype(_DT[dt]).itemsize), dt, (nv, 3), k, normal_format, 0))
                elif a["codec"] == CODEC_COLOR:
                    oc = color_components or a["components"]
                    plan.append((i, a["name"], take(nv * oc), "u8", (nv, oc), k, FMT_UINT8, oc))
                else:
                    plan.append((i, a["name"], take(nv * a["components"] * 4), "f32", (nv, a["components"]), k, FMT_FLOAT, 0))
            if nf and (only is None or "index" in only):
                dt = "u16" if index16 else "u32"
                plan.append((i, "index", take(nf * 3 * np.dtype(_DT[dt]).itemsize), dt, (nf, 3), -1, FMT_UINT16 if index16 else FMT_UINT32, 0))
        total = max(off, 256)
        buf = torch.empty(total, dtype=torch.uint8, device=dev) if fill is None else torch.full((total,), fill, dtype=torch.uint8, device=dev)
        _torch_ready(dev)                            # (the fill runs on torch's stream; the decode on the context's own)
        base = buf.data_ptr()
        nattr_total = sum(info.nattr for info in self.infos)
        binds = (AttrBinding * max(nattr_total, 1))()
        first = np.cumsum([0] + [info.nattr for info in self.infos])
        index_ptrs = (C.c_void_p * max(len(self), 1))()
        index_fmt = np.full(max(len(self), 1), FMT_UINT32, dtype=np.uint32)
        outs = [dict() for _ in self.infos]
        tdt = {"f32": torch.float32, "i16": torch.int16, "u8": torch.uint8, "u32": torch.int32, "u16": torch.int16}
        for (i, name, o, dt, shape, slot, fmt, oc) in plan:
            nbytes = int(np.prod(shape)) * np.dtype(_DT[dt]).itemsize
            view = buf[o:o + nbytes].view(tdt[dt]).view(*shape) if nbytes else torch.empty(shape, dtype=tdt[dt], device=dev)
            outs[i][name] = (view, dt)
            if slot >= 0:
                bd = binds[int(first[i]) + slot]
                bd.buffer = base + o; bd.format = fmt; bd.out_components = oc
            else:
                index_ptrs[i] = base + o; index_fmt[i] = fmt
        self._keep = (buf, binds, index_ptrs, index_fmt)
        self._interleaved = None
        self.outputs = outs
        self.rebind()
        return outs

    def allocate_interleaved(self, normal_format=FMT_INT16, index16=True, fill: Optional[int] = None):
        """Render-ready outputs: every blob gets ONE interleaved vertex buffer - each vertex a record of
        position f32x3 | normal (i16x3 + 2 bytes of padding, or f32x3) | uv f32x2 | colour u8x4 | other generic attributes f32xN, in
        that order, for the attributes the blob has - bound through crthip_attr_binding.stride, and a u16 (or u32) index buffer.
        host_outputs() de-interleaves.  Returns the per-blob record layouts {name: (offset, dtype, components)} and strides."""
        import torch
        dev = torch.device("cuda", self.ctx.device)
        order = {"position": 0, "normal": 1, "uv": 2, "color": 3}
        off = 0
        metas = []
        for i, info in enumerate(self.infos):
            nv, nf = info.nvert, info.nface
            attrs = info.attrs()
            rec, layout = 0, {}
            for k, a in sorted(enumerate(attrs), key=lambda ka: (order.get(ka[1]["name"], 4), ka[0])):
                if a["codec"] == CODEC_NORMAL:
                    dt, comps, size, fmt, oc = ("i16", 3, 8, FMT_INT16, 0) if normal_format == FMT_INT16 else ("f32", 3, 12, FMT_FLOAT, 0)
                elif a["codec"] == CODEC_COLOR:
                    dt, comps, size, fmt, oc = "u8", 4, 4, FMT_UINT8, 4
                else:
                    dt, comps, size, fmt, oc = "f32", a["components"], 4 * a["components"], FMT_FLOAT, 0
                layout[a["name"]] = (rec, dt, comps, k, fmt, oc)
                rec += size
            off = (off + 255) & ~255
            vb = off; off += nv * rec
            ib = None
            if nf:
                off = (off + 255) & ~255
                ib = off; off += nf * 3 * (2 if index16 and nv < 65536 else 4)
            metas.append((vb, rec, layout, ib, bool(nf and index16 and nv < 65536)))
        total = max(off, 256)
        buf = torch.empty(total, dtype=torch.uint8, device=dev) if fill is None else torch.full((total,), fill, dtype=torch.uint8, device=dev)
        _torch_ready(dev)                            # (the fill runs on torch's stream; the decode on the context's own)
        base = buf.data_ptr()
        nattr_total = sum(info.nattr for info in self.infos)
        binds = (AttrBinding * max(nattr_total, 1))()
        first = np.cumsum([0] + [info.nattr for info in self.infos])
        index_ptrs = (C.c_void_p * max(len(self), 1))()
        index_fmt = np.full(max(len(self), 1), FMT_UINT32, dtype=np.uint32)
        for i, (vb, rec, layout, ib, i16) in enumerate(metas):
            for name, (o, dt, comps, k, fmt, oc) in layout.items():
                bd = binds[int(first[i]) + k]
                bd.buffer = base + vb + o; bd.format = fmt; bd.out_components = oc; bd.stride = rec
            if ib is not None:
                index_ptrs[i] = base + ib; index_fmt[i] = FMT_UINT16 if i16 else FMT_UINT32
        self._keep = (buf, binds, index_ptrs, index_fmt)
        self._interleaved = metas
        self.outputs = [dict() for _ in self.infos]
        self.rebind()
        return metas

    def rebind(self):
        """(Re)apply the bindings prepared by allocate_outputs: one C call for the whole batch."""
        buf, binds, index_ptrs, index_fmt = self._keep
        _check(lib().crthip_batch_bind_all(self.handle, binds, index_ptrs, _np_ptr(index_fmt)))

    def bind(self, i: int, bindings: Sequence[AttrBinding], index_ptr: Optional[int] = None, index_format=FMT_UINT32):
        arr = (AttrBinding * max(len(bindings), 1))(*bindings)
        _check(lib().crthip_batch_bind(self.handle, i, arr, C.c_void_p(index_ptr) if index_ptr else None, index_format))

    def host_outputs(self, i: int) -> Dict[str, np.ndarray]:
        """Copy blob i's outputs to the host as numpy arrays with the reference's dtypes."""
        res = {}
        metas = getattr(self, "_interleaved", None)
        if metas is not None and not self.outputs[i]:
            vb, rec, layout, ib, i16 = metas[i]
            nv, nf = self.infos[i].nvert, self.infos[i].nface
            raw = self._keep[0][vb:vb + nv * rec].cpu().numpy().reshape(nv, rec) if nv * rec else np.zeros((nv, rec), np.uint8)
            for name, (o, dt, comps, k, fmt, oc) in layout.items():
                w = comps * np.dtype(_DT[dt]).itemsize
                res[name] = np.ascontiguousarray(raw[:, o:o + w]).view(_DT[dt]).reshape(nv, comps)
            if ib is not None:
                nb = nf * 3 * (2 if i16 else 4)
                res["index"] = self._keep[0][ib:ib + nb].cpu().numpy().view(np.uint16 if i16 else np.uint32).reshape(nf, 3)
            res["nvert"], res["nface"] = nv, nf
            return res
        for name, (t, dt) in self.outputs[i].items():
            a = t.cpu().numpy()
            res[name] = a.view(_DT[dt]) if a.dtype != _DT[dt] else a
        res["nvert"], res["nface"] = self.infos[i].nvert, self.infos[i].nface
        return res

    # -- run ---------------------------------------------------------------------------------------
    def decode(self):
        _check(lib().crthip_batch_decode(self.handle))

    def sync(self, raise_on_error=True) -> np.ndarray:
        st = np.zeros(max(len(self), 1), dtype=np.int32)
        code = lib().crthip_batch_sync(self.handle, _np_ptr(st))
        if code != 0 and raise_on_error:
            _check(code)
        return st[:len(self)]

    def stats(self) -> BatchStats:
        s = BatchStats()
        _check(lib().crthip_batch_get_stats(self.handle, C.byref(s)))
        return s

    def kernel_times(self) -> Dict[str, dict]:
        t = KernelTimes()
        _check(lib().crthip_batch_kernel_times(self.handle, C.byref(t)))
        return t.as_dict()

    def debug_read(self, i: int, what: str, nbytes: int) -> np.ndarray:
        out = np.zeros(max(nbytes, 1), dtype=np.uint8)
        n = lib().crthip_batch_debug_read(self.handle, i, what.encode(), _np_ptr(out), nbytes)
        if n < 0:
            _check(int(n))
        return out[:n]

    def close(self):
        if self.handle:
            lib().crthip_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """crthip_pool: the multi-GPU decode pool - `devices` GPUs, `threads` host threads per GPU with `depth` batches in flight each,
    one shared work queue over the submitted batches, no collective (include/corto_hip.h)."""

    def __init__(self, devices: Sequence[int], threads: int = 2, depth: int = 3):
        self.devices = list(devices)
        self.handle = C.c_void_p()
        arr = np.array(self.devices, dtype=np.int32)
        _check(lib().crthip_pool_create(len(self.devices), _np_ptr(arr), threads, depth, C.byref(self.handle)))
        self.lanes = int(lib().crthip_pool_lanes(self.handle))
        self.warning = lib().crthip_pool_warning(self.handle).decode()
        self._keep = None

    def device_cpus(self, slot: int) -> List[int]:
        """host CPUs of pool device `slot`'s NUMA node (what its worker threads are pinned to); [] when sysfs names none"""
        buf = np.zeros(4096, dtype=np.int32)
        L = lib()
        L.crthip_pool_device_cpus.restype = C.c_int64
        L.crthip_pool_device_cpus.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        n = int(L.crthip_pool_device_cpus(self.handle, slot, _np_ptr(buf), buf.size))
        if n < 0:
            _check(n)
        return [int(x) for x in buf[:min(n, buf.size)]]

    def set_packed_host_blobs(self, on: bool = True):
        """items run without device arenas whose blobs are views of ONE pinned host buffer (pinned_host_arena) go up straight from it"""
        _check(lib().crthip_pool_set_packed_host_blobs(self.handle, int(on)))

    def set_outputs_to_host(self, on: bool = True):
        """every step ends with a D2H copy of its outputs into the lane's pinned host block (SURVEY 8d's secondary region); lane_read then
        returns what that copy delivered"""
        _check(lib().crthip_pool_set_outputs_to_host(self.handle, int(on)))

    def set_render_layouts(self, on: bool = True):
        """int16 normals and uint16 indices (blobs of fewer than 65 536 vertices) in every lane's outputs: SURVEY 8f3's layouts, 22 % fewer output bytes a C4 blob"""
        _check(lib().crthip_pool_set_render_layouts(self.handle, int(on)))

    def run(self, items, steps: int, warmup: int = 0, arenas=None):
        """items: list of batches (each a list of aligned uint8 blobs).  arenas: None (every step uploads its blobs) or, per item, a
        list with one device tensor per pool device (the item's blobs resident there in arena_layout order).
        Returns (PoolReport, completion times of the timed steps in seconds since the timed region began)."""
        n = len(items)
        arr = (PoolItem * n)()
        keep = []
        for j, blobs in enumerate(items):
            ptrs = (C.c_void_p * max(len(blobs), 1))(*[b.ctypes.data for b in blobs])
            lens = np.array([len(b) for b in blobs], dtype=np.uint32)
            ar = None
            if arenas is not None and arenas[j] is not None:
                ar = (C.c_void_p * len(self.devices))(*[(t.data_ptr() if t is not None else None) for t in arenas[j]])
            keep.append((ptrs, lens, ar))
            arr[j].nblobs = len(blobs); arr[j].blobs = C.cast(ptrs, C.c_void_p); arr[j].lens = lens.ctypes.data
            arr[j].device_arena = C.cast(ar, C.c_void_p) if ar is not None else None
        self._keep = (arr, keep, items, arenas)
        rep = PoolReport()
        stamps = np.zeros(max(steps, 1), dtype=np.float64)
        _check(lib().crthip_pool_run(self.handle, n, arr, steps, warmup, C.byref(rep), _np_ptr(stamps)))
        return rep, stamps[:steps]

    def lane_item(self, lane: int):
        slot = C.c_uint32()
        it = int(lib().crthip_pool_lane_item(self.handle, lane, C.byref(slot)))
        return it, int(slot.value)

    def lane_read(self, lane: int, blob: int, what: str, dtype, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        n = int(lib().crthip_pool_lane_read(self.handle, lane, blob, what.encode(), _np_ptr(out), out.nbytes))
        if n < 0:
            _check(n)
        assert n == out.nbytes, (n, out.nbytes)
        return out

    def close(self):
        if self.handle:
            lib().crthip_pool_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_ready(device) -> None:
    """The library's HIP streams are non-blocking: they do not wait for work torch queued on ITS stream (a fill of the output block, an
    upload).  Whoever prepares device buffers with torch finishes that work before handing the pointers over (include/corto_hip.h:
    "device buffers")."""
    import torch
    torch.cuda.current_stream(device).synchronize()


def upload_arena(blobs: Sequence[np.ndarray], device: int = 0):
    """Stage blobs back to back (16-byte aligned starts) into one device tensor: the 'inputs resident in
    HBM' form that Batch(device_arena=...) consumes."""
    import torch
    offs, total = arena_layout([len(b) for b in blobs])
    host = np.zeros(max(total, 16), dtype=np.uint8)
    for b, o in zip(blobs, offs):
        host[int(o):int(o) + len(b)] = b
    arena = torch.from_numpy(host).to(torch.device("cuda", device))
    _torch_ready(arena.device)
    return arena


def pinned_host_arena(blobs: Sequence[np.ndarray]):
    """The blobs back to back (crthip_arena_layout) in ONE pinned host buffer: returns (the pinned torch tensor - keep it alive -, the
    list of numpy views of the blobs inside it).  Handed to a context / pool with set_packed_host_blobs(True), the views are uploaded
    with one DMA copy straight from the buffer.

    REUSE HAZARD: the copy is only enqueued when a batch is created / reset - the library neither snapshots the buffer nor waits for the
    copy.  The buffer must stay alive AND UNCHANGED until the batch that reads it has been synced (Batch.sync / the pool run's return);
    writing the next batch's blobs into the same buffer before that corrupts the decode in flight silently.  Rotate as many buffers as
    there are batches in flight.  $CORTO_HIP_CHECK_PINNED=1 makes the library verify (hipPointerGetAttributes) that a buffer handed over
    this way really is pinned host memory and refuse it (CRTHIP_E_ARGUMENT) otherwise."""
    import torch
    offs, total = arena_layout([len(b) for b in blobs])
    pin = torch.zeros(max(total, 16), dtype=torch.uint8).pin_memory()
    host = pin.numpy()
    views = []
    for b, o in zip(blobs, offs):
        host[int(o):int(o) + len(b)] = b
        views.append(host[int(o):int(o) + len(b)])
    return pin, views


class Decoder:
    """Mirror of crt::Decoder for one blob with HOST (numpy) output buffers; decode() runs on the GPU through
    crthip_decode_host.  Same member names as upstream (include/corto/decoder.h:38-73)."""

    _default_ctx: Dict[int, Context] = {}

    def __init__(self, blob: np.ndarray, device: int = 0):
        self.blob = blob
        self.info = probe(blob)
        self.nvert, self.nface = self.info.nvert, self.info.nface
        self.exif = probe_exif(blob)
        self.data = {a["name"]: a for a in self.info.attrs()}
        self._bind = {}
        self._index = None
        self._index_format = FMT_UINT32
        self.device = device

    def hasAttr(self, name: str) -> bool:
        return name in self.data

    def setAttribute(self, name: str, buffer: np.ndarray, fmt: int, out_components: int = 0) -> bool:
        if name not in self.data:
            return False
        self._bind[name] = (buffer, fmt, out_components)
        return True

    def setPositions(self, buffer: np.ndarray) -> bool:
        return self.setAttribute("position", buffer, FMT_FLOAT)

    def setNormals(self, buffer: np.ndarray) -> bool:
        return self.setAttribute("normal", buffer, FMT_INT16 if buffer.dtype == np.int16 else FMT_FLOAT)

    def setUvs(self, buffer: np.ndarray) -> bool:
        return self.setAttribute("uv", buffer, FMT_FLOAT)

    def setColors(self, buffer: np.ndarray, components: int = 4) -> bool:
        return self.setAttribute("color", buffer, FMT_UINT8, components)

    def setIndex(self, buffer: np.ndarray):
        self._index = buffer
        self._index_format = FMT_UINT16 if buffer.dtype == np.uint16 else FMT_UINT32

    def decode(self):
        ctx = Decoder._default_ctx.get(self.device)
        if ctx is None:
            ctx = Decoder._default_ctx[self.device] = Context(self.device)
        attrs = self.info.attrs()
        binds = (AttrBinding * max(len(attrs), 1))()
        for k, a in enumerate(attrs):
            if a["name"] in self._bind:
                buf, fmt, oc = self._bind[a["name"]]
                binds[k].buffer = buf.ctypes.data; binds[k].format = fmt; binds[k].out_components = oc
        idx = C.c_void_p(self._index.ctypes.data) if self._index is not None else None
        _check(lib().crthip_decode_host(ctx.handle, _np_ptr(self.blob), len(self.blob), binds, idx, self._index_format))


def tunstall_decode_blocks(ctx: Context, host_blocks: np.ndarray, device_blocks, block_offsets, device_out, out_offsets):
    """Stand-alone Tunstall decode of device-resident blocks (HBM-roofline run). Returns per-kernel ms."""
    bo = np.ascontiguousarray(block_offsets, dtype=np.uint64)
    oo = np.ascontiguousarray(out_offsets, dtype=np.uint64)
    t = KernelTimes()
    _torch_ready(device_out.device)                   # device_blocks / device_out are the caller's torch tensors
    _check(lib().crthip_tunstall_decode_blocks(ctx.handle, len(bo), _np_ptr(host_blocks), C.c_void_p(device_blocks.data_ptr()),
                                               _np_ptr(bo), C.c_void_p(device_out.data_ptr()), _np_ptr(oo), C.byref(t)))
    return t.as_dict()


def tunstall_encode_blocks(ctx: Context, streams: Sequence[np.ndarray], with_times: bool = False):
    """GPU encoder stage (crthip_tunstall_encode_blocks): the reference's OutStream::tunstall_compress block of every
    byte stream in `streams`.  Returns a list of uint8 arrays (one block each) [, per-kernel ms]."""
    streams = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
    n = len(streams)
    ptrs = (C.c_void_p * max(n, 1))(*[s.ctypes.data if len(s) else None for s in streams])
    sizes = np.array([len(s) for s in streams], dtype=np.uint32)
    cap = int(sizes.sum()) + 600 * n + 64
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    t = KernelTimes()
    r = lib().crthip_tunstall_encode_blocks(ctx.handle, n, ptrs, _np_ptr(sizes), _np_ptr(out), cap, _np_ptr(offs), C.byref(t))
    if r < 0:
        _check(int(r))
    blocks = [out[int(offs[i]):int(offs[i + 1])].copy() for i in range(n)]
    return (blocks, t.as_dict()) if with_times else blocks


ENC_SYMBOLS, ENC_ARRAY, ENC_VALUES_I32, ENC_VALUES_I8 = 0, 1, 2, 3


class EncStream(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("count", C.c_uint32), ("components", C.c_uint32), ("reserved", C.c_uint32), ("values", C.c_void_p)]


def encode_values(ctx: Context, streams, entropy: int = 1, with_times: bool = False):
    """GPU encoder stage (crthip_encode_values).  streams: list of (kind, array); ARRAY / VALUES arrays are (count, N) int32
    (int8 for ENC_VALUES_I8), SYMBOLS arrays are uint8.  Returns one uint8 array per stream: "u32 nwords | words | blocks"."""
    n = len(streams)
    desc = (EncStream * max(n, 1))()
    keep = []
    cap = 1024
    for i, (kind, a) in enumerate(streams):
        a = np.ascontiguousarray(a, dtype=np.uint8 if kind == ENC_SYMBOLS else np.int8 if kind == ENC_VALUES_I8 else np.int32)
        keep.append(a)
        count = a.shape[0] if a.ndim else 0
        comps = 1 if kind == ENC_SYMBOLS or a.ndim < 2 else a.shape[1]
        desc[i].kind, desc[i].count, desc[i].components = kind, count, comps
        desc[i].values = a.ctypes.data if a.size else None
        cap += a.size * 5 + 600 * (comps + 1) + 64
    out = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    t = KernelTimes()
    r = lib().crthip_encode_values(ctx.handle, entropy, n, desc, _np_ptr(out), cap, _np_ptr(offs), C.byref(t))
    if r < 0:
        _check(int(r))
    res = [out[int(offs[i]):int(offs[i + 1])].copy() for i in range(n)]
    return (res, t.as_dict()) if with_times else res
