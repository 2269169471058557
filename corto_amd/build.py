"""Build libcorto_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m corto_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: an FMA contraction changes the
bits of the reference's BORDER normals; IEEE divide/sqrt are hipcc's default
(-fhip-fp32-correctly-rounded-divide-sqrt).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("CORTO_BUILD_LIBDIR") or os.path.join(HERE, "lib")   # (probes: a variant library beside the product, tools/ab_build.sh)
LIB = os.path.join(LIBDIR, "libcorto_hip.so")
VENEER = os.path.join(LIBDIR, "libcortocodec_hip.so")
EMVENEER = os.path.join(LIBDIR, "libcorto_em_hip.so")    # upstream's wasm/JS C ABI (include/corto/emcorto.h) over the facade
CLI = os.path.join(LIBDIR, "corto_hip")                   # the `corto` command line tool on this repo's encoder + GPU decoder (tools/corto_hip_cli.cpp)   # legacy Unity C ABI (include/corto/corto_codec.h) over the facade
SOURCES = ["k_tunstall.hip", "k_stream.hip", "k_mesh.hip", "k_delta.hip", "k_normal.hip", "k_encode.hip", "batch.cpp", "plan_carve.cpp", "plan_jobs.cpp", "plan_group.cpp", "plan_launch.cpp", "crt_format.cpp", "decoder_facade.cpp", "encoder.cpp", "encode_gpu.cpp", "pool.cpp"]
# every header of csrc/ is a dependency of every object (a stale object behind a changed header decodes with yesterday's kernel)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [
    os.path.join("..", "..", "include", "corto_hip.h"), os.path.join("..", "..", "include", "corto", "decoder.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-x", "hip"] + ["-D" + d for d in os.environ.get("CORTO_BUILD_DEFINES", "").split(",") if d]   # (probes: CORTO_TUN_STAMPS)


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc()] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    vsrc = os.path.join(CSRC, "unity_veneer.cpp")
    inc = os.path.join(HERE, "..", "include")
    if force or _stale(VENEER, [vsrc, LIB, os.path.join(inc, "corto", "corto_codec.h"), os.path.join(inc, "corto", "decoder.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", inc, vsrc, "-o", VENEER,
               "-L", LIBDIR, "-lcorto_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    esrc = os.path.join(CSRC, "em_veneer.cpp")
    if force or _stale(EMVENEER, [esrc, LIB, os.path.join(inc, "corto", "emcorto.h"), os.path.join(inc, "corto", "decoder.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", inc, esrc, "-o", EMVENEER,
               "-L", LIBDIR, "-lcorto_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    csrc = os.path.join(HERE, "..", "tools", "corto_hip_cli.cpp")
    if os.path.exists(csrc) and (force or _stale(CLI, [csrc, LIB, os.path.join(inc, "corto_hip.h"), os.path.join(inc, "corto", "decoder.h")])):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", inc, csrc, "-o", CLI, "-L", LIBDIR, "-lcorto_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
