#!/usr/bin/env python3
"""bench.py — decode throughput of the MI355X hot path on BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config C4 of BASELINE.json): a batch of 256 independent ~4K-triangle .crt blobs
(2 112 verts / 4 096 tris each; position 14 bit + uv 12 bit + normal 10 bit BORDER + rgba 6/7/6/5)
= 1 048 576 triangles / 540 672 vertices PER GPU (weak scaling: config C5 = 8 GPUs x 256 blobs).
A "step" = one pass of the hot path over that batch on SURVEY.md 8d's PRIMARY timed region: the compressed blobs start in pinned HOST memory
(one buffer per batch, arena layout) and the decoded outputs end in device HBM - upload of the 3.7 MB of .crt bytes over PCIe, re-plan
(crthip_batch_reset: host walk of every blob) + bind + decode (descriptor upload, all kernels) + sync.  (Rounds 1-3 quoted the rate with the
compressed inputs already resident in HBM as `value`; VERDICT r3 asked for 8d's primary region instead.  The resident-input rate - no
PCIe inside the step - is measured in the same run and reported beside it as `resident_inputs`, with its own regions and roofline.)
Steps run on the library's decode pool (crthip_pool, csrc/pool.cpp): per GPU --host-threads (default 5) native host threads
each keep --depth (default 4) batches in flight, every batch on its own context (own HIP streams, scratch and output
block), all threads of all GPUs pulling batches from ONE work queue (an atomic counter) - no collective anywhere.
Timing: barrier + device sync, then W warm-up steps flow straight into the K timed steps (the pipeline is NOT drained in
between); the clock runs from the completion of the last warm-up step to the completion of the K-th timed step, a few more
steps are queued behind so that every context is still busy when the clock stops, then everything drains, barrier + sync.
So `--steps 20` measures the same steady state as `--steps 480`.  value = K batches / that time (max over ranks).
With K < 100 the K-step region is timed R = ~2000/K times back to back (no drain in between) and the MEDIAN region is reported, every
region's time in `timed_regions`: completions of sixteen batches in flight come in bursts, and one region of 20 steps lands
anywhere within -15 / +30 % of the long-run rate (tools/pool_probe.py).
`python bench.py --gpus N` without a launcher runs the N GPUs from this one process (one pool, shared queue, N x K steps);
under torch.distributed.run every rank runs its own one-GPU pool on its shard (seeds 256*rank ..) and RCCL carries only the
barrier and the max-over-ranks of the time.
`single_batch` reports the unpipelined latency of one step; the per-kernel HIP-event times come from that phase.
One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")   # one HW queue per HIP stream of the pipelined contexts (the ROCm default of 4 makes streams
                                                   # share queues, and a shared queue serialises its kernels).  Round 4: 20 contexts on 20 queues
                                                   # (5 x 4; 4 x 5 and 11 x 2 on 22 are the same) are 3 % faster than 16 on 16; 24 is slower again
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NBLOBS = 256


def hbm_ceiling(torch):
    """what plain device kernels reach on this GPU: torch's fill (write-only) and copy (read + write) over 1 GiB - the measured
    STREAM-like ceiling SURVEY 8d asks to be quoted beside the 8 TB/s spec"""
    x = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
    def t(f, n=8):
        f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    nb = x.numel() * 4
    out = {"fill_GBps": round(nb / t(lambda: x.zero_()) / 1e6, 0), "copy_GBps": round(2 * nb / t(lambda: y.copy_(x)) / 1e6, 0),
           "note": "torch zero_() / copy_() over 1 GiB on this box (write-only, read+write)"}
    del x, y
    return out


def pcie_ceiling(torch, h2d_bytes, d2h_bytes):
    """what plain DMA copies of a step's sizes reach on this box: pinned host -> HBM for the compressed bytes of one step, HBM -> pinned host for
    its outputs; back to back on one stream, HIP events (the ceilings the `pcie` and `secondary_region` fractions are quoted against)"""
    def t(dst, src, n=24):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            dst.copy_(src, non_blocking=True)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = {}
    for key, nb, up in (("h2d", h2d_bytes, True), ("d2h", d2h_bytes, False)):
        nb = max(int(nb), 4096)
        h = torch.empty(nb, dtype=torch.uint8, pin_memory=True); d = torch.empty(nb, dtype=torch.uint8, device="cuda")
        ms = t(d, h) if up else t(h, d)
        out[key + "_GBps"] = round(nb / ms / 1e6, 2); out[key + "_bytes"] = nb
        if up:
            h16 = torch.empty(16 << 20, dtype=torch.uint8, pin_memory=True); d16 = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
            out["h2d_16MB_GBps"] = round((16 << 20) / t(d16, h16) / 1e6, 2)
            del h16, d16
        del h, d
    out["note"] = "torch copy_(non_blocking) between pinned host memory and HBM, 24 copies of a step's size back to back on one stream, HIP events"
    return out


def first_iteration(blobs_path_seed=0):
    """'report first-iteration separately' - a PROCESS that has never touched the GPU creates a context, plans the C4 batch, allocates, uploads and
    decodes it once: run in a child process so that this process's loaded code objects, scratch pools and clocks do not flatter it.  Times exclude the
    interpreter's imports (python, numpy, torch: seconds, and not this library's)."""
    import subprocess
    code = (
        "import time, json, sys, numpy as np\n"
        "import torch, corto_amd as ca\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "blobs, z = bench.load_blobs(first_seed=0)\n"
        "t0 = time.perf_counter(); ctx = ca.Context(0); t1 = time.perf_counter()\n"
        "b = ca.Batch(ctx, blobs); b.allocate_outputs(); t2 = time.perf_counter()\n"
        "b.decode(); st = b.sync(); t3 = time.perf_counter()\n"
        "assert (st == 0).all()\n"
        "b.decode(); b.sync(); t4 = time.perf_counter()\n"
        "b2 = ca.Batch(ctx, blobs); b2.allocate_outputs(); b2.decode(); b2.sync(); t5 = time.perf_counter()\n"
        "print(json.dumps({'context_create_ms': round((t1 - t0) * 1e3, 2), 'plan_allocate_upload_ms': round((t2 - t1) * 1e3, 2), 'first_decode_ms': round((t3 - t2) * 1e3, 2),"
        " 'first_iteration_ms': round((t3 - t0) * 1e3, 2), 'second_decode_ms': round((t4 - t3) * 1e3, 3), 'second_batch_ms': round((t5 - t4) * 1e3, 3)}))\n"
    ) % os.path.dirname(os.path.abspath(__file__))
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.abspath(__file__)))
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": (out.stderr or out.stdout)[-300:]}
        r = json.loads(line[-1])
        r["note"] = ("a fresh process: crthip_ctx_create (HIP runtime + device init, streams, events), then plan + output allocation + H2D of the C4 batch from host memory, then the first "
                     "decode (code objects loaded, scratch pools allocated, clocks cold) to its sync; second_decode_ms: the same batch object again; second_batch_ms: a new batch on the warm context")
        return r
    except Exception as e:                                  # noqa: BLE001
        return {"error": repr(e)[:300]}


def gpu_numa_cpus(torch, ndev):
    """the host CPUs next to each visible GPU (PCI bus id -> sysfs numa_node -> cpulist), [] where sysfs does not say: what crthip_pool pins its
    threads to, known here BEFORE any pool exists so that the threads per GPU can be sized for the GPUs that share a socket"""
    out = []
    for i in range(ndev):
        cpus = []
        try:
            pr = torch.cuda.get_device_properties(i)
            bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read().strip())
            if node >= 0:
                for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus += list(range(int(a), int(b or a) + 1))
        except Exception:                                       # noqa: BLE001  (no sysfs entry, an older torch: unknown)
            cpus = []
        out.append(cpus)
    return out


def window_stats(stamps, lanes, nwin=20):
    """best and median ms per step over up to `nwin` consecutive windows of the timed steps' completion times (a window is at least
    four rounds of the pool's contexts: completions come in bursts of about one per context)"""
    k = len(stamps)
    nwin = max(1, min(nwin, k // max(4 * lanes, 1)))
    edges = [round(i * k / nwin) for i in range(nwin + 1)]
    t = np.concatenate([[0.0], np.asarray(stamps)])
    per = [(t[edges[i + 1]] - t[edges[i]]) / (edges[i + 1] - edges[i]) * 1e3 for i in range(nwin) if edges[i + 1] > edges[i]]
    return {"windows": len(per), "best_ms_per_step": round(float(min(per)), 4), "median_ms_per_step": round(float(np.median(per)), 4)}


def facade_per_blob(ca, blobs, device):
    """What a crt::Decoder caller pays per blob (the drop-in use): crthip_decode_host - upload, decode, one download into
    the caller's host buffers - one blob after the other on one thread, and on four threads with a context each."""
    import ctypes as C
    import threading
    L = ca.lib()
    sample = blobs[:16]
    def make(blob):
        info = ca.probe(blob)
        outs = {"position": np.zeros((info.nvert, 3), np.float32), "normal": np.zeros((info.nvert, 3), np.float32),
                "color": np.zeros((info.nvert, 4), np.uint8), "uv": np.zeros((info.nvert, 2), np.float32)}
        idx = np.zeros((info.nface, 3), np.uint32)
        attrs = info.attrs()
        binds = (ca.AttrBinding * len(attrs))()
        for k, a in enumerate(attrs):
            binds[k].buffer = outs[a["name"]].ctypes.data
            binds[k].format = ca.FMT_UINT8 if a["name"] == "color" else ca.FMT_FLOAT
            binds[k].out_components = 4 if a["name"] == "color" else 0
        return outs, idx, binds
    def loop(ctx, reps, out_t, jobs):
        t0 = time.perf_counter()
        for _ in range(reps):
            for blob, (outs, idx, binds) in jobs:
                ca._check(L.crthip_decode_host(ctx.handle, blob.ctypes.data, len(blob), binds, idx.ctypes.data, ca.FMT_UINT32))
        out_t.append((time.perf_counter() - t0) / (reps * len(jobs)))
    ctxs = [ca.Context(device) for _ in range(4)]
    jobs = [[(b, make(b)) for b in sample] for _ in range(4)]
    t = []
    loop(ctxs[0], 2, t, jobs[0])
    t = []
    loop(ctxs[0], 12, t, jobs[0])
    one = t[0]
    from oracle import oracle as oc
    ref = oc.decode(sample[3])
    got, idx, _ = jobs[0][3][1]
    for k in ("position", "normal", "color", "uv"):
        assert got[k].tobytes() == ref[k].tobytes(), ("facade leg: bit-exact check failed", k)
    assert idx.tobytes() == ref["index"].tobytes()
    ts = []
    ths = [threading.Thread(target=loop, args=(ctxs[i], 12, ts, jobs[i])) for i in range(4)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    wall = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    # the C++ facade itself (crt::Decoder, decoder_facade.cpp): its decode() calls are coalesced across threads
    cpp = {}
    try:
        import subprocess, tempfile
        from corto_amd import build as bld
        d = tempfile.mkdtemp(prefix="corto_facade_")
        exe = os.path.join(d, "facade_threads")
        subprocess.check_call(["g++", "-O2", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_threads.cpp"),
                               "-L", bld.LIBDIR, "-lcorto_hip", "-Wl,-rpath," + bld.LIBDIR, "-o", exe])
        files = []
        for i, b in enumerate(sample):
            f = os.path.join(d, "b%02d.crt" % i); np.asarray(b).tofile(f); files.append(f)
        for nt in (1, 4, 16):
            out = subprocess.run([exe, str(nt), "24" if nt == 1 else "16", "-", *files], capture_output=True, text=True, timeout=120, env=dict(os.environ, CORTO_HIP_DEVICE=str(device)))
            if out.returncode == 0:
                kv = dict(l.split() for l in out.stdout.strip().splitlines() if len(l.split()) == 2)
                cpp["%d_threads" % nt] = {"us_per_blob": float(kv["wall_us_per_blob"]), "decode_call_us": float(kv["per_decode_us"])}
        cpp["note"] = ("crt::Decoder::decode() on N threads, a fresh Decoder per blob (tests/cpp/facade_threads.cpp): us_per_blob = wall time over all decodes; concurrent calls are "
                       "coalesced into one batch by the facade's combiner.  A blob's decode is one serial chain on the GPU (~0.3 ms whatever else runs), so N synchronous callers "
                       "cannot get below (0.3 ms + host) / N per blob")
    except Exception as e:                           # (no compiler on the box, ...: the ctypes numbers stand)
        cpp["error"] = str(e)[:200]
    return {"one_thread_us": round(one * 1e6, 1), "four_threads_us_per_blob": round(wall / (4 * 12 * len(sample)) * 1e6, 1), "crt_decoder_cpp": cpp,
            "mtri_per_s_one_thread": round(4096 / one / 1e6, 2),
            "note": "crthip_decode_host per C4-unit blob (host .crt -> host arrays: what crt::Decoder::decode() costs), contexts reused; "
                    "timed through ctypes (a few us of Python per call included).  Beside it: cpu_baseline's per-blob time = 4096 / (Mtri/s)"}


def load_blobs(first_seed=0):
    """256 distinct C4-unit blobs: corto_amd.synth.bumpy_sphere(64, 32, seed) (2 112 verts / 4 096 tris) encoded
    by the repo's own .crt writer (csrc/encoder.cpp, byte-identical to the reference encoder; tests/test_encoder_cpu.py).
    Host-side input synthesis, outside the timed region."""
    import corto_amd as ca
    from corto_amd import synth
    blobs = [ca.encode(synth.bumpy_sphere(64, 32, seed=first_seed + i), position_bits=14, uv_bits=12, normal_bits=10,
                       normal_prediction=ca.BORDER) for i in range(NBLOBS)]
    z = np.load(os.path.join(ROOT, "tests", "golden", "c4_blobs16.npz"))
    return blobs, z


def cpu_baseline(blobs, budget_s=12.0):
    """Reference (oracle/_ref) or the C restatement (port) timed on ONE host core over a bounded sample."""
    from oracle import refcodec as rc
    sample = blobs[:16]
    tris = sum(4096 for _ in sample)
    if rc.available():
        kind = "reference"
        def run():
            t = 0
            for b in sample:
                ns, _ = rc.decode_timed(b, 1)
                t += int(ns[0])
            return t * 1e-9
    else:
        from oracle import oracle as oc
        kind = "port"
        def run():
            t0 = time.perf_counter()
            for b in sample:
                oc.decode(b)
            return time.perf_counter() - t0
    run()
    best, n, t_start = 1e9, 0, time.perf_counter()
    while time.perf_counter() - t_start < budget_s:
        best = min(best, run()); n += 1
    verts = 2112 * len(sample)
    out = {"value": round(tris / best / 1e6, 3), "unit": "Mtri/s", "mverts_per_s": round(verts / best / 1e6, 3), "cores": 1, "kind": kind,
           "sample": "16 distinct C4-unit blobs (65 536 tris) decoded back to back, best of %d passes in %.0f s; ctor+set*+decode per blob" % (n, budget_s),
           "host_cpus": os.cpu_count()}
    if rc.available():
        # context: the same decoder on every host core at once, one blob per thread (the C call releases the GIL)
        nthr = os.cpu_count() or 1
        t0 = time.perf_counter()
        ndone = rc.decode_mt(sample, nthr, 4.0)
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(ndone * 4096 / wall / 1e6, 1), "unit": "Mtri/s", "threads": nthr,
                            "note": "reference decoder on %d C++ threads, one blob per thread, for %.1f s (context, not the baseline)" % (nthr, wall)}
    return out


def sources_sha256():
    """one hash over everything libcorto_hip.so is built from (corto_amd/csrc/*, the public headers, the build flags): what ties a
    committed PMC profile to the kernels that are running now.  tools/prof_run.sh stamps it into the profile it writes."""
    import hashlib
    from corto_amd import build as b
    h = hashlib.sha256()
    files = sorted(os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC) if f.endswith((".hip", ".cpp", ".h")))
    files += [os.path.join(ROOT, "include", "corto_hip.h"), os.path.join(ROOT, "include", "corto", "decoder.h")]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    h.update(" ".join(b.FLAGS).encode())
    return h.hexdigest()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command (profiles/, separate
    --pmc runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  Returns (bytes, note): bytes is None when no
    profile is committed OR when the newest one was taken from other sources than the ones this library was built from (the
    profile carries the hash of corto_amd/csrc at the time, sources_sha256()) - a number from another kernel is not reported."""
    import glob
    name = {"topology_lds": "corto_hip::k_topology_lds", "topology": "corto_hip::k_topology", "delta_mesh": "corto_hip::k_delta_mesh",
            "tunstall_tables": "corto_hip::k_tun_tables", "tunstall_decode": "corto_hip::k_tun_decode", "tunstall_stream": "corto_hip::k_tun_stream",
            "delta_lds16": "corto_hip::k_delta_lds16", "unpack_wave": "corto_hip::k_unpack_wave", "unpack_extract": "corto_hip::k_unpack_extract",
            "normal_blob": "corto_hip::k_normal_blob"}.get(kernel)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_per_dispatch.json")))
    if not name or not files:
        return None, "no PMC profile committed for this kernel"
    prof = json.load(open(files[-1]))
    tag = os.path.basename(files[-1])
    stamp = prof.get("_sources_sha256")
    if stamp != sources_sha256():
        return None, "%s was taken from other sources (%s) than this build (%s): not reported" % (tag, (stamp or "unstamped")[:12], sources_sha256()[:12])
    d = prof.get(name, {})
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None, "%s has no FETCH_SIZE / WRITE_SIZE for %s" % (tag, name)
    return int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024), "%s (sources %s = this build): (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch" % (tag, stamp[:12])


def tunstall_scaled(ctx, ca, z, table_ids=None):
    """The Tunstall kernels on a run large enough to leave the launch-latency regime:
    64 streams x 4 Mi codewords with the dictionaries of real log streams, random codeword payloads."""
    import torch
    from oracle import oracle as oc
    rng = np.random.default_rng(1)
    kat = np.load(os.path.join(ROOT, "tests", "golden", "tunstall_kat.npz"))
    nstream, ncode = 64, 4 << 20
    blocks, sizes = [], []
    for i in range(nstream):
        pr = kat["probs_%02d" % (table_ids[i % len(table_ids)] if table_ids else 7 + (i % 50))]
        _, ln, _ = oc.tunstall_tables(pr)
        payload = rng.integers(0, 256, ncode, dtype=np.uint8)
        size = int(ln[payload].sum(dtype=np.int64))
        hdr = bytes([len(pr)]) + pr.tobytes() + size.to_bytes(4, "little") + ncode.to_bytes(4, "little")
        blocks.append(np.frombuffer(hdr + payload.tobytes(), dtype=np.uint8)); sizes.append(size)
    offs, total = [], 0
    for b in blocks:
        offs.append(total); total += (len(b) + 15) & ~15
    host = np.zeros(total + 16, dtype=np.uint8)
    for b, o in zip(blocks, offs):
        host[o:o + len(b)] = b
    oo, ot = [], 0
    for s in sizes:
        oo.append(ot); ot += (s + 15) & ~15
    ctx.set_profiling(True)                      # per-kernel HIP events (crthip_kernel_times)
    dblk = torch.from_numpy(host).cuda()
    dout = torch.empty(ot + 16, dtype=torch.uint8, device="cuda")
    best = None
    ca.tunstall_decode_blocks(ctx, host, dblk, offs, dout, oo)             # first use: scratch allocation, code upload
    runs = []
    for _ in range(9):                                                     # the run whose kernels together took least; every run's total beside it
        t = ca.tunstall_decode_blocks(ctx, host, dblk, offs, dout, oo)
        runs.append(round(sum(v["ms"] for v in t.values()), 4))
        if best is None or sum(v["ms"] for v in t.values()) < sum(v["ms"] for v in best.values()):
            best = t
    rd, wr = nstream * ncode, sum(sizes)
    dec_ms = best["tunstall_decode"]["ms"]
    all_ms = sum(v["ms"] for v in best.values())
    return {"streams": nstream, "codewords_per_stream": ncode, "bytes_read": rd, "bytes_written": wr,
            "decode_kernel_ms": round(dec_ms, 4), "all_tunstall_kernels_ms": round(all_ms, 4), "all_tunstall_kernels_ms_runs": sorted(runs),
            "kernels_ms": {k: round(v["ms"], 4) for k, v in best.items()},
            "all_kernels_GBps": round((rd + wr) / all_ms / 1e6, 1), "all_kernels_frac_of_8TBps": round((rd + wr) / all_ms / 1e6 / 8000.0, 4),
            "decode_kernel_GBps": round((rd + wr) / dec_ms / 1e6, 1), "read_only_GBps": round(rd / dec_ms / 1e6, 1),
            "frac_of_8TBps": round((rd + wr) / dec_ms / 1e6 / 8000.0, 4)}


def other_configs(ctx, ca):
    """BASELINE.json's single-object configs, one decode each (parity cases in tests/; timed here for the record): they do not take
    the many-blobs route: a single mesh is one serial CLERS chain."""
    from corto_amd import synth
    out = {}
    for key, mesh, kw in (("C2_mesh_128k_verts", synth.bumpy_sphere(512, 250, seed=1), dict(normal_prediction=ca.BORDER)),
                          ("C3_cloud_167k_points", synth.point_cloud(578, 289, seed=2), dict(normal_prediction=ca.DIFF))):
        blob = ca.encode(mesh, position_bits=14, uv_bits=12, normal_bits=10, **kw)
        b = ca.Batch(ctx, [blob]); b.allocate_outputs()
        b.decode(); b.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            b.decode(); b.sync()
        dt = (time.perf_counter() - t0) / 3
        out[key] = {"ms": round(dt * 1e3, 3), "mverts_per_s": round(mesh.nvert / dt / 1e6, 2)}
        if mesh.nface:
            out[key]["mtri_per_s"] = round(mesh.nface / dt / 1e6, 2)
        b.close()
        try:                                         # the reference decoder on one host core, same blob (oracle/_ref, when it travelled)
            from oracle import refcodec as rc
            if rc.available():
                ns, _ = rc.decode_timed(blob, 3)
                out[key]["cpu_reference_ms"] = round(float(min(ns)) * 1e-6, 3)
        except Exception:
            pass
    # where a single mesh stops losing to one CPU core: the same decode on grids of growing size
    sweep = []
    # (round 6: + the 33-66K-vertex meshes with 16-bit positions, which rounds 1-5 sent through the stretch walk over L2: 30K vertices is the LDS records' size,
    # 50K the first beyond them - `ns_per_vertex` should fall, not jump, from one to the other)
    for nu, nv, pbits in ((64, 32, 14), (128, 64, 14), (240, 124, 16), (256, 125, 14), (320, 160, 16), (384, 190, 14)):
        mesh = synth.bumpy_sphere(nu, nv, seed=1)
        blob = ca.encode(mesh, position_bits=pbits, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
        b = ca.Batch(ctx, [blob]); b.allocate_outputs()
        b.decode(); b.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            b.decode(); b.sync()
        dt = (time.perf_counter() - t0) / 5
        row = {"triangles": int(mesh.nface), "vertices": int(mesh.nvert), "position_bits": pbits, "gpu_ms": round(dt * 1e3, 3), "ns_per_vertex": round(dt * 1e9 / mesh.nvert, 1)}
        b.close()
        try:
            from oracle import refcodec as rc
            if rc.available():
                ns, _ = rc.decode_timed(blob, 3)
                row["cpu_reference_ms"] = round(float(min(ns)) * 1e-6, 3)
        except Exception:
            pass
        sweep.append(row)
    out["single_mesh_sweep"] = sweep
    out["note"] = ("one object per decode: no blob-level parallelism.  The 128K-vertex mesh is ONE serial CLERS chain, whose (VERTEX LEFT) runs the whole wave "
                   "does 63 pairs at a time (DESIGN.md 3.1); single_mesh_sweep shows the size below which one mesh alone is faster on a CPU core "
                   "(a decode is a dozen dependent launches and one serial chain whatever the size) - batches are the GPU's case")
    return out


def encoder_stage(ctx, ca):
    """SURVEY.md 8f-4, measured: the GPU Tunstall coder (crthip_tunstall_encode_blocks) on the entropy-coder load of one C4 batch -
    2 304 streams of 2 112 bit-width logs - beside the reference's OutStream::tunstall_compress on one host core (oracle/_ref)."""
    rng = np.random.default_rng(4)
    streams = [np.clip(np.rint(rng.normal(2 + (k % 9), 0.6 + 0.1 * (k % 7), 2112)), 0, 31).astype(np.uint8) for k in range(2304)]
    nbytes = sum(len(s) for s in streams)
    ca.tunstall_encode_blocks(ctx, streams[:64])
    best, times = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        blocks, tk = ca.tunstall_encode_blocks(ctx, streams, with_times=True)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, times = dt, tk
    out = {"workload": "2304 streams x 2112 symbols (bit-width logs of one C4 batch)", "symbols": nbytes, "compressed_bytes": int(sum(len(b) for b in blocks)),
           "call_ms": round(best * 1e3, 3), "msymbols_per_s": round(nbytes / best / 1e6, 1),
           "kernel_ms": {k: round(v["ms"], 4) for k, v in times.items()},
           "note": "call = upload + device histogram, probabilities (std::sort's order), dictionaries, tries and parse + download + framing; enc_trie.launches = streams whose tables the host had to make; blocks byte-identical to the reference's"}
    # whole blobs: crthip_encode (host) vs crthip_encode_gpu (value coding + entropy coder on the device), same bytes
    from corto_amd import synth
    out["blob_encode"] = {}
    for key, mesh in (("C4_unit_4k_tris", synth.bumpy_sphere(64, 32, seed=1)), ("C2_mesh_128k_verts", synth.bumpy_sphere(512, 250, seed=1))):
        kw = dict(position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER)
        ca.encode(mesh, ctx=ctx, **kw)
        t0 = time.perf_counter(); a = ca.encode(mesh, **kw); t_host = time.perf_counter() - t0
        t0 = time.perf_counter(); b = ca.encode(mesh, ctx=ctx, **kw); t_gpu = time.perf_counter() - t0
        out["blob_encode"][key] = {"host_ms": round(t_host * 1e3, 3), "gpu_stages_ms": round(t_gpu * 1e3, 3), "identical": bool(a.tobytes() == b.tobytes()),
                                   "crt_bytes": int(len(a))}
    out["blob_encode"]["note"] = "the CLERS encode and the prediction deltas stay on the host in both: one object at a time the device stages (quantisation, value coding, entropy coder) do not pay for their transfers and syncs - they are for batches of streams (above)"
    try:
        from oracle import refcodec as rc
        if rc.available():
            t0 = time.perf_counter()
            for s in streams[:256]:
                rc.tunstall_compress_block(s)
            dt = (time.perf_counter() - t0) * 9
            out["cpu_reference_ms"] = round(dt * 1e3, 3)
            out["cpu_reference_note"] = "OutStream::tunstall_compress (oracle/_ref) on one host core, 256 of the streams timed, scaled to 2304"
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--depth", type=int, default=4, help="batches in flight (contexts) per host thread; 1 = unpipelined")
    ap.add_argument("--host-threads", type=int, default=5, help="native host threads per GPU feeding it (crthip_pool)")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the `sustained` leg (one long timed region on the same pool); 0 skips it")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--all-legs", action="store_true", help="N > 1: also run the single-GPU characterisation legs (CPU baseline, Tunstall at roofline scale, single objects, irregular batches) that belong to the N = 1 line")
    ap.add_argument("--no-tunstall-scaled", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the single-object C2 / C3 decodes (tools/prof_run.sh: keeps the rocprofv3 kernel averages about the C4 batch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import corto_amd as ca
    # test hook (tools/, tests): BENCH_SHARE_GPU=1 lets several ranks / pool devices share one GPU (gloo for the barrier: RCCL
    # refuses duplicate devices) - exercises the N>1 code paths on a 1-GPU box; never set by the driver
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (the MI355X path has no CPU fallback)")
    if world > 1:
        # one process per GPU (torch.distributed.run): this rank's pool has one device
        mode = "process-per-gpu"
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        if share:
            local_rank = local_rank % ndev
        elif local_rank >= ndev:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, ndev))
        devices = [local_rank]
        slots_global = [rank]                      # which shard of C5 each pool device's "home" item is
        n_gpus = world
    else:
        # one process, N GPUs: one pool over all of them, one shared work queue
        mode = "single-process-queue" if args.gpus > 1 else "single-gpu"
        if args.gpus > ndev and not share:
            raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, ndev))
        devices = [i % ndev for i in range(args.gpus)]
        slots_global = list(range(args.gpus))
        n_gpus = args.gpus
    torch.cuda.set_device(devices[0])
    dist = None
    red_dev = torch.device("cuda", devices[0])
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world); red_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=red_dev)

    # the single-GPU characterisations - reference decoder on the host's cores, Tunstall at roofline scale, the single-object configs, the
    # drop-in class, irregular / realistic batches - belong to the N = 1 line (the task: `cpu_baseline` on rank 0 at N = 1 only); an N > 1
    # run is the scaling measurement and skips them unless asked (--all-legs): the other ranks would only wait at a barrier meanwhile
    from corto_amd import shard
    n1_only_skipped = False
    if n_gpus > 1 and not args.all_legs:
        args.no_cpu = args.no_tunstall_scaled = args.no_other_configs = True
        n1_only_skipped = True
    # C5 = n_gpus x 256 blobs cut into contiguous work-balanced ranges (one work item each); seeds 256*g .. 256*g+255 for range g
    ranges = shard.balanced_ranges([4096 + 2112] * (NBLOBS * n_gpus), n_gpus)
    z = None
    items = []
    for g in slots_global:
        lo, hi = ranges[g]
        assert hi - lo == NBLOBS
        blobs_g, z = load_blobs(first_seed=lo)
        items.append(blobs_g)
    blobs = items[0]
    depth, nthreads = max(1, args.depth), max(1, args.host_threads)
    # host threads per GPU, sized for the GPUs that share a NUMA node (8 GPUs on two sockets: 4 x threads feeder threads a socket) and for this
    # process' cpuset - a pool that time-shares its feeder threads measures the host: said loudly (stderr and `config.host_threads_note`) when a GPU cannot get a core of its own
    numa = gpu_numa_cpus(torch, ndev)
    phys = sorted(set(range(n_gpus if not share else ndev)) & set(range(ndev))) if world > 1 else sorted(set(devices))
    plan_t, threads_note = shard.plan_host_threads(nthreads, [numa[d] for d in phys], sorted(os.sched_getaffinity(0)))
    if min(plan_t) < nthreads or threads_note:
        nthreads = min(plan_t)
        print("bench.py: " + threads_note, file=sys.stderr)
    # compressed inputs resident in HBM before the timed region: item j on ITS pool device (j % N: crthip_pool's home-shard-first policy) -
    # sharding, not replication; a device asked to decode another's item would upload it inside the step
    arenas = [[ca.upload_arena(it, d) if k == j % len(devices) else None for k, d in enumerate(devices)] for j, it in enumerate(items)]
    pool = ca.Pool(devices, threads=nthreads, depth=depth)

    ctx = ca.Context(devices[0])
    arena = arenas[0][0]
    b0 = ca.Batch(ctx, blobs, device_arena=arena)
    b0.allocate_outputs()
    stats0 = None
    # one step = plan + bind + decode + sync, through the C ABI only
    import ctypes as C
    L = ca.lib()
    n = len(blobs)
    ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in blobs])
    lens = np.array([len(x) for x in blobs], dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    handle = [None]

    def launch(from_host=False):
        buf, binds, index_ptrs, index_fmt = b0._keep
        dev = None if from_host else C.c_void_p(arena.data_ptr())
        if handle[0] is None:
            h = C.c_void_p()
            ca._check(L.crthip_batch_create(ctx.handle, n, ptrs, lens.ctypes.data_as(C.c_void_p), dev, C.byref(h)))
            handle[0] = h
        else:
            ca._check(L.crthip_batch_reset(handle[0], n, ptrs, lens.ctypes.data_as(C.c_void_p), dev))
        h = handle[0]
        ca._check(L.crthip_batch_bind_all(h, binds, index_ptrs, index_fmt.ctypes.data_as(C.c_void_p)))
        ca._check(L.crthip_batch_decode(h))
        return h

    def finish(h, st=status):
        ca._check(L.crthip_batch_sync(h, st.ctypes.data_as(C.c_void_p)))
        assert (st == 0).all(), st

    def device_sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        ctx.sync()

    def barrier():
        shard.barrier(dist, device_sync)

    # ---- unpipelined phase: latency of one batch, and per-kernel device times (HIP events around every kernel, on the
    # stream the kernels run on) with nothing else on the GPU - the numbers the rocprofv3 summary in profiles/ agrees with
    ctx.set_profiling(True)
    kt_acc = {}
    solo_steps = max(3, min(10, args.steps))
    finish(launch())
    barrier()
    t0 = time.perf_counter()
    for _ in range(solo_steps):
        h = launch()
        finish(h)
        kt = ca.KernelTimes()
        L.crthip_batch_kernel_times(h, C.byref(kt))
        for k, v in kt.as_dict().items():
            a = kt_acc.setdefault(k, [0.0, 0]); a[0] += v["ms"]; a[1] += v["launches"]
        st = ca.BatchStats(); L.crthip_batch_get_stats(h, C.byref(st)); stats0 = st
    solo_ms = (time.perf_counter() - t0) / solo_steps * 1e3
    ctx.set_profiling(False)
    # PCIe-inclusive variant of the same unpipelined step: the blobs start in host memory and crthip_batch_create uploads them
    t0 = time.perf_counter()
    for _ in range(solo_steps):
        finish(launch(from_host=True))
    h2d_ms = (time.perf_counter() - t0) / solo_steps * 1e3
    # ... and with the decoded outputs copied back to (pinned) host memory as well: what a host-side caller of crt::Decoder pays
    dbuf = b0._keep[0]
    hbuf = torch.empty(dbuf.shape, dtype=dbuf.dtype, pin_memory=True)
    t0 = time.perf_counter()
    for _ in range(solo_steps):
        finish(launch())
        hbuf.copy_(dbuf, non_blocking=True); torch.cuda.synchronize()
    d2h_ms = (time.perf_counter() - t0) / solo_steps * 1e3

    # ---- the timed region (see the module docstring): the pool runs W warm-up steps straight into exactly K timed steps per GPU
    nloc = len(devices)
    # SURVEY 8d's primary region: every item's blobs in ONE pinned host buffer (arena layout), uploaded over PCIe inside every step
    # ... allocated (and so first touched and pinned) on a CPU of the GPU's own NUMA node: with eight GPUs on two sockets a buffer on the
    # far socket's memory is read across the inter-socket link on every step
    pins = []
    numa_local_inputs = 0
    for j, it in enumerate(items):
        cpus = pool.device_cpus(j % nloc)
        old_aff = os.sched_getaffinity(0)
        moved = False
        if cpus:
            try:
                os.sched_setaffinity(0, set(cpus) & old_aff or set(cpus)); moved = True
            except OSError:
                pass
        try:
            pins.append(ca.pinned_host_arena(it))
        finally:
            if moved:
                os.sched_setaffinity(0, old_aff); numa_local_inputs += 1
    host_items = [v for _, v in pins]
    pool.set_packed_host_blobs(True)
    # whatever W is: every context used (scratch pools are allocated on first use) and the GPU at its working clocks before the clock starts
    prewarm_steps = int(os.environ.get("BENCH_PREWARM_STEPS", "0")) or 8 * pool.lanes
    pool.run(host_items, steps=prewarm_steps * nloc, warmup=0, arenas=None)
    # ... and for at least 0.3 s (round 6: on a box that had just been idle - or profiled - the first ~100 steps behind 8 rounds of the contexts still ran at
    # 0.10-0.21 ms: clocks and the host's page / NUMA state, not the pipeline's depth)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < float(os.environ.get("BENCH_PREWARM_SECONDS", "0.3")):
        pool.run(host_items, steps=16 * pool.lanes * nloc, warmup=0, arenas=None)
    barrier()
    # A K-step region is timed R times back to back (no drain in between) and the MEDIAN region is the one reported: with sixteen batches
    # in flight completions come in bursts, and a single region of K = 20 steps (1.25 rounds of the contexts) lands anywhere within
    # -15 / +30 % of the long-run rate (tools/pool_probe.py); every region's time is in `timed_regions`.  K >= 100: five regions (one 480-step region that meets a multi-ms stall reads 25 % low).
    R = 5 if args.steps >= 100 else max(5, -(-2000 // args.steps) | 1)     # (an odd number of regions, ~2 000 steps in all = 0.18 s: round 3's five regions' median still moved +-8 % run to run, round 4's 25 +-4 % - 11.0 .. 12.1 Gtri/s on five boxes where the 480-step form read 11.9 .. 12.2)
    # (the W warm-up steps the caller asked for, plus eight rounds of the pool's contexts (round 6: two rounds left the first regions at 0.10-0.21 ms on the driver's box): with twenty batches in flight a pipeline that was empty
    # when the run began is not full after five steps, and the first regions would time its filling)
    ramp = 8 * pool.lanes
    rep, stamps = pool.run(host_items, steps=R * args.steps * nloc, warmup=args.warmup * nloc + ramp, arenas=None)
    barrier()
    kk = args.steps * nloc
    tt = np.concatenate([[0.0], np.asarray(stamps, dtype=np.float64)])             # completion times of the timed steps, from the last warm-up step's
    regions = [shard.max_over_ranks(float(tt[(j + 1) * kk] - tt[j * kk]), dist, red_dev) for j in range(R)]
    elapsed = float(np.median(regions))
    elapsed_mean = shard.max_over_ranks(rep.elapsed_s, dist, red_dev) / R          # (all R regions together: a stall anywhere shows here)
    if rep.failed_blobs or rep.first_error:
        raise SystemExit("bench.py: %d blobs failed to decode (first status %d)" % (rep.failed_blobs, rep.first_error))
    if rep.devices_used != nloc:
        raise SystemExit("bench.py: only %d of %d pool devices decoded anything: %s" % (rep.devices_used, nloc, list(rep.steps_per_device)[:nloc]))
    steps_per_device = list(rep.steps_per_device)[:nloc]
    # every context's LAST step started from an output block the pool had just filled with 0xA5 (pool.cpp): the check below cannot pass on
    # bytes a warm-up step left behind
    if rep.poisoned_lanes != pool.lanes:
        raise SystemExit("bench.py: only %d of %d contexts ended on a poisoned output block" % (rep.poisoned_lanes, pool.lanes))

    # untimed bit-exactness check: what EVERY context of EVERY device decoded last, sampled, against the CPU oracle ...
    import hashlib
    from oracle import oracle as oc
    dts = {"position": (np.float32, 3), "normal": (np.float32, 3), "color": (np.uint8, 4), "uv": (np.float32, 2), "index": (np.uint32, 3)}
    checked = 0
    for lane in range(pool.lanes):
        it, slot = pool.lane_item(lane)
        assert it >= 0, ("a pool context never ran", lane)
        for i in range(lane % 17, NBLOBS, 17 * 3):
            ref = oc.decode(items[it][i])
            for k, (dt, w) in dts.items():
                cnt = (ref["nface"] if k == "index" else ref["nvert"]) * w
                got = pool.lane_read(lane, i, k, dt, cnt)
                assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed", lane, slot, it, i, k)
            checked += 1
        tail = pool.lane_read(lane, 0, "#tail", np.uint8, 256)       # behind every output array: still the poison
        assert (tail == 0xA5).all(), ("output block was not poisoned before the lane's last step", lane)
    for i in range(0, NBLOBS, 17):                # ... the unpipelined context's outputs as well ...
        got, ref = b0.host_outputs(i), oc.decode(blobs[i])
        for k in ("position", "normal", "color", "uv", "index"):
            assert got[k].tobytes() == ref[k].tobytes(), ("bit-exact check failed", i, k)
    if rank == 0:
        for i in range(16):                       # ... and seeds 0-15 against the digests of the reference decoder's own output
            got = b0.host_outputs(i)
            for k in ("position", "normal", "color", "uv", "index"):
                d = hashlib.sha256(np.ascontiguousarray(got[k]).tobytes()).hexdigest()
                assert d == z["%s_sha256_%02d" % (k, i)].tobytes().decode(), ("bit-exact check failed (golden)", i, k)

    # beside `value`: the same pipelined steps with the compressed inputs ALREADY RESIDENT in HBM (what rounds 1-3 quoted as `value`): no PCIe
    # inside the step.  Same R regions of K steps, median reported.
    # (the side legs run 600 steps whatever K is: a 120-step region of sixteen batches in flight lands anywhere within -40 / +5 % of the
    # long-run rate - round 3, profiles/EXPERIMENTS.md - and 600 steps of these take 0.1-0.15 s)
    fh_steps = 600 * nloc
    pool.set_packed_host_blobs(False)
    pool.run(items, steps=4 * pool.lanes * nloc, warmup=0, arenas=arenas)
    barrier()
    rep_res, stamps_res = pool.run(items, steps=R * args.steps * nloc, warmup=args.warmup * nloc + ramp, arenas=arenas)
    barrier()
    tr = np.concatenate([[0.0], np.asarray(stamps_res, dtype=np.float64)])
    regions_res = [shard.max_over_ranks(float(tr[(j + 1) * kk] - tr[j * kk]), dist, red_dev) for j in range(R)]
    elapsed_res = float(np.median(regions_res))
    tris_res = shard.sum_over_ranks(float(rep_res.triangles), dist, red_dev) / R
    if rep_res.failed_blobs:
        raise SystemExit("bench.py: resident-input leg: %d failed blobs" % rep_res.failed_blobs)
    # ... and with the blobs scattered over PAGEABLE host memory (256 separate numpy arrays): the worker thread gathers them into a pinned
    # image first (3.7 MB of memcpy per step on the host thread)
    fhh_steps = 2000 * nloc       # (the from-host legs run longer: round 3 met an occasional 7-8 ms stall - tools/fromhost_ab.py - that a 60 ms leg cannot average out)
    pool.run(items, steps=4 * pool.lanes * nloc, warmup=0, arenas=None)      # (untimed: every feeder's pinned image of an arena is allocated on first use)
    barrier()
    rep_h, stamps_h = pool.run(items, steps=fhh_steps, warmup=2 * pool.lanes, arenas=None)
    barrier()
    elapsed_h = shard.max_over_ranks(rep_h.elapsed_s, dist, red_dev)
    pool.set_packed_host_blobs(True)

    # sustained: the same pool, the same steps, for at least --sustain seconds in ONE region (steady clocks and thermals; what a 2 ms region cannot show)
    sustained = None
    if args.sustain > 0:
        est = max(elapsed / args.steps, 1e-6)
        n_sus = int(min(max(args.sustain / est * 1.3, 200), 400000)) * nloc
        barrier()
        rep_s, stamps_s = pool.run(host_items, steps=n_sus, warmup=2 * pool.lanes, arenas=None)
        barrier()
        el_s = shard.max_over_ranks(rep_s.elapsed_s, dist, red_dev)
        if el_s < args.sustain:                                                  # (the estimate came from a slow region: once more, sized by what this leg itself ran at)
            n_sus = int(min(n_sus * args.sustain / max(el_s, 1e-6) * 1.25, 400000 * nloc))
            rep_s, stamps_s = pool.run(host_items, steps=n_sus, warmup=2 * pool.lanes, arenas=None)
            barrier()
            el_s = shard.max_over_ranks(rep_s.elapsed_s, dist, red_dev)
        tris_s = shard.sum_over_ranks(float(rep_s.triangles), dist, red_dev)
        if rep_s.failed_blobs or rep_s.poisoned_lanes != pool.lanes:
            raise SystemExit("bench.py: sustained leg: %d failed blobs, %d of %d contexts poisoned" % (rep_s.failed_blobs, rep_s.poisoned_lanes, pool.lanes))
        for lane in range(0, pool.lanes, 3):                                     # ... and what it left is the oracle's bytes too
            it, _slot = pool.lane_item(lane)
            ref = oc.decode(items[it][11 + lane])
            for k, (dt, w) in dts.items():
                got = pool.lane_read(lane, 11 + lane, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed (sustained)", lane, k)
        # windows of ~0.1 s
        ts = np.concatenate([[0.0], np.asarray(stamps_s, dtype=np.float64)])
        nwin = max(1, int(ts[-1] / 0.1))
        edges = [round(i * (len(ts) - 1) / nwin) for i in range(nwin + 1)]
        per = [(ts[edges[i + 1]] - ts[edges[i]]) / (edges[i + 1] - edges[i]) * 1e3 for i in range(nwin) if edges[i + 1] > edges[i]]
        sustained = {"seconds": round(el_s, 3), "steps": n_sus // nloc, "mtri_per_s": round(tris_s / el_s / 1e6, 2), "ms_per_step": round(el_s / (n_sus / nloc) * 1e3, 4),
                     "windows": len(per), "best_window_ms_per_step": round(float(min(per)), 4), "median_window_ms_per_step": round(float(np.median(per)), 4),
                     "worst_window_ms_per_step": round(float(max(per)), 4), "host_us_per_step_per_thread": round(float(rep_s.host_us_per_step), 1),
                     "note": "one timed region of >= %.1f s on the same pool and region as `value` (same items, uploaded from pinned host memory inside every step), windows of ~0.1 s; outputs poisoned before the last round and checked against the oracle" % args.sustain}

    # beside `value`: the same pipelined steps with one Tunstall dictionary built PER STREAM ($CORTO_TUN_SHARE=2; read when a context is
    # made, so: a second pool).  By default the streams of a batch that carry the same probability table share one dictionary, and the
    # synthetic blobs - one generator, 256 seeds, the same connectivity - repeat tables far more than unrelated meshes would.
    # (the main pool is closed first: its sixteen contexts' streams would share the sixteen hardware queues with the side pools' - round 3's
    # first side-leg numbers were taken that way and came out 10-30 % under what the same pool does alone, tools/shape_probe.py)
    pool_lanes, pool_warning = pool.lanes, pool.warning
    pool.close()
    os.environ["CORTO_TUN_SHARE"] = "2"            # one dictionary per stream whatever repeats (still two kernels: dictionaries, then decodes)
    pool_ns = ca.Pool(devices, threads=nthreads, depth=depth)
    del os.environ["CORTO_TUN_SHARE"]
    pool_ns.set_packed_host_blobs(True)
    pool_ns.run(host_items, steps=4 * pool_ns.lanes, warmup=0, arenas=None)
    barrier()
    rep_ns, _ = pool_ns.run(host_items, steps=fh_steps, warmup=2 * pool_ns.lanes, arenas=None)
    barrier()
    elapsed_ns = shard.max_over_ranks(rep_ns.elapsed_s, dist, red_dev)
    tris_ns = shard.sum_over_ranks(float(rep_ns.triangles), dist, red_dev)
    pool_ns.close()

    # beside `value` too: the same batch shape with IRREGULAR connectivity (every grid quad's diagonal flipped per seed: valences 4-8, no two
    # blobs share a CLERS stream; shorter (VERTEX LEFT) runs, shorter scan blocks) - what the pipeline does when meshes are not lat-long grids
    irregular = None
    realistic = None
    if rank == 0 and not args.no_other_configs:
        from corto_amd import synth
        iblobs = [ca.encode(synth.bumpy_sphere_flipped(64, 32, seed=i), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER) for i in range(NBLOBS)]
        pin_i, iviews = ca.pinned_host_arena(iblobs)
        pool_i = ca.Pool(devices[:1], threads=nthreads, depth=depth)
        pool_i.set_packed_host_blobs(True)
        pool_i.run([iviews], steps=4 * pool_i.lanes, warmup=0, arenas=None)
        rep_i, st_i = pool_i.run([iviews], steps=fh_steps // nloc, warmup=2 * pool_i.lanes, arenas=None)
        for lane in range(0, pool_i.lanes, 5):                                  # bit-exact spot check against the oracle
            i = 7 * lane + 3
            ref = oc.decode(iblobs[i])
            for k, (dt, w) in dts.items():
                got = pool_i.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed (irregular)", lane, i, k)
        irregular = {"mtri_per_s": round(rep_i.triangles / rep_i.elapsed_s / 1e6, 2), "ms_per_step": round(rep_i.elapsed_s / (fh_steps // nloc) * 1e3, 4),
                     "steps": fh_steps // nloc, "topology_fallbacks": int(rep_i.topology_fallbacks), "failed_blobs": int(rep_i.failed_blobs),
                     "note": "one GPU, same timed region as `value` (pinned host -> HBM); 256 x bumpy_sphere_flipped(64, 32, seed): 2112 verts / 4096 tris each, every quad's diagonal flipped with probability 1/2"}
        pool_i.close()
        # `realistic`: everything the headline's best case leaves out, at once - irregular connectivity, one dictionary PER STREAM (no two
        # blobs of unrelated meshes share tables), and the compressed blobs uploaded from host memory inside every step (SURVEY 8d's primary region)
        # Round 5: the meshes of this leg are DELAUNAY discs with holes (synth.delaunay_disc: no lattice anywhere, valence 1-11, a rim and 6-10 more boundary
        # loops a blob, ~3 900 triangles - the C4 unit's size); rounds 2-4 ran it on grids with flipped diagonals (still `irregular_connectivity` above)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
            rblobs = list(ex.map(lambda sd: ca.encode(synth.delaunay_disc(2310, seed=sd, holes=6 + sd % 5), position_bits=14, uv_bits=12, normal_bits=10, normal_prediction=ca.BORDER), range(NBLOBS)))
        pin_r, rviews = ca.pinned_host_arena(rblobs)
        os.environ["CORTO_TUN_SHARE"] = "2"
        pool_r = ca.Pool(devices[:1], threads=nthreads, depth=depth)
        del os.environ["CORTO_TUN_SHARE"]
        pool_r.run([rblobs], steps=4 * pool_r.lanes, warmup=0, arenas=None)
        r_steps = max(fh_steps // nloc, 200)
        rep_r, st_r = pool_r.run([rblobs], steps=r_steps, warmup=2 * pool_r.lanes, arenas=None)
        assert rep_r.poisoned_lanes == pool_r.lanes
        for lane in range(1, pool_r.lanes, 4):
            for i in (5 * lane + 2, NBLOBS - 1 - lane):
                ref = oc.decode(ca.aligned_blob(rblobs[i]))
                for k, (dt, w) in dts.items():
                    got = pool_r.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                    assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed (realistic)", lane, i, k)
        pool_r.set_packed_host_blobs(True)
        rep_rp, _ = pool_r.run([rviews], steps=r_steps, warmup=2 * pool_r.lanes, arenas=None)
        pool_r.set_packed_host_blobs(False)
        realistic = {"mtri_per_s": round(rep_r.triangles / rep_r.elapsed_s / 1e6, 2), "mverts_per_s": round(rep_r.vertices / rep_r.elapsed_s / 1e6, 2),
                     "ms_per_step": round(rep_r.elapsed_s / r_steps * 1e3, 4), "steps": r_steps,
                     "triangles_per_step": int(sum(ca.probe(x).nface for x in rblobs)), "vertices_per_step": int(sum(ca.probe(x).nvert for x in rblobs)),
                     "topology_fallbacks": int(rep_r.topology_fallbacks), "failed_blobs": int(rep_r.failed_blobs), **window_stats(st_r, pool_r.lanes),
                     "h2d_bytes_per_step": int(sum(((len(x) + 15) & ~15) for x in rblobs)),
                     "packed_pinned_mtri_per_s": round(rep_rp.triangles / rep_rp.elapsed_s / 1e6, 2) if not rep_rp.failed_blobs else None,
                     "note": "one GPU; 256 Delaunay discs with 6-10 holes each (synth.delaunay_disc(2310, seed, holes): no lattice, valence 1-11, 7-11 boundary loops a blob) + $CORTO_TUN_SHARE=2 (one dictionary "
                             "per stream) + compressed blobs scattered over pageable HOST memory, gathered and uploaded over PCIe inside every step (packed_pinned_mtri_per_s: from one pinned buffer, the region "
                             "of `value`); outputs poisoned before the last round, bit-exact spot check against the oracle.  The number to expect from unrelated scanned meshes; `value` is the best case"}
        pool_r.close()

    # SURVEY 8d's SECONDARY region, pipelined: the timed region of `value` plus the D2H copy of every step's decoded outputs into pinned host memory
    # (queued behind the step's kernels on its context's stream; a step is complete when its copy is) - what a host-side consumer of the outputs sees,
    # i.e. the reference's own region (decode() into host buffers, src/main.cpp:266-300).  Its own pool (20 pinned mirrors of 32 MB), the main one closed.
    secondary = None
    secondary_render = None
    pcie_caps = None
    if rank == 0 and not args.no_other_configs:
        out_bytes_step = int(stats0.output_bytes)
        pcie_caps = pcie_ceiling(torch, int(stats0.arena_bytes), out_bytes_step)
        pool_s = ca.Pool(devices[:1], threads=nthreads, depth=depth)
        pool_s.set_packed_host_blobs(True)
        pool_s.set_outputs_to_host(True)
        s_steps = 400
        pool_s.run(host_items[:1], steps=2 * pool_s.lanes, warmup=0, arenas=None)
        rep_s2, st_s2 = pool_s.run(host_items[:1], steps=s_steps, warmup=2 * pool_s.lanes, arenas=None)
        assert rep_s2.poisoned_lanes == pool_s.lanes and not rep_s2.failed_blobs
        for lane in range(0, pool_s.lanes, 3):                                   # what the D2H copies delivered is the oracle's bytes
            for i in (lane, NBLOBS - 1 - lane):
                ref = oc.decode(blobs[i])
                for k, (dt, w) in dts.items():
                    got = pool_s.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                    assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed (secondary region, host copy)", lane, i, k)
        ms_s = rep_s2.elapsed_s / s_steps * 1e3
        d2h_rate = out_bytes_step / (ms_s * 1e-3) / 1e9
        secondary = {"mtri_per_s": round(rep_s2.triangles / rep_s2.elapsed_s / 1e6, 2), "mverts_per_s": round(rep_s2.vertices / rep_s2.elapsed_s / 1e6, 2), "ms_per_step": round(ms_s, 4),
                     "steps": s_steps, **window_stats(st_s2, pool_s.lanes), "d2h_bytes_per_step": out_bytes_step, "d2h_GBps": round(d2h_rate, 2),
                     "d2h_measured_ceiling_GBps": pcie_caps["d2h_GBps"], "pcie_frac": round(d2h_rate / pcie_caps["d2h_GBps"], 4),
                     "note": "SURVEY 8d secondary region, pipelined: pinned-host .crt -> HBM (as `value`) -> decoded outputs in pinned HOST memory, one D2H copy a step behind its kernels on the "
                             "context's stream, %d batches in flight; bit-exact check on the host copies; PCIe-bound on the way back (outputs are ~9x the compressed bytes)" % pool_s.lanes}
        # the same region with SURVEY 8f3's render layouts (int16 normals, uint16 index): what a renderer binds as vertex / index buffers - fewer bytes over PCIe
        pool_s.set_render_layouts(True)
        pool_s.run(host_items[:1], steps=2 * pool_s.lanes, warmup=0, arenas=None)
        rep_s3, st_s3 = pool_s.run(host_items[:1], steps=s_steps, warmup=2 * pool_s.lanes, arenas=None)
        assert rep_s3.poisoned_lanes == pool_s.lanes and not rep_s3.failed_blobs
        dts_r = dict(dts); dts_r["normal"] = (np.int16, 3); dts_r["index"] = (np.uint16, 3)
        out_bytes_r = 0
        for lane in range(0, pool_s.lanes, 3):
            for i in (lane, NBLOBS - 1 - lane):
                ref = oc.decode(blobs[i], normal_format=oc.FMT_INT16, index16=True)
                for k, (dt, w) in dts_r.items():
                    got = pool_s.lane_read(lane, i, k, dt, (ref["nface"] if k == "index" else ref["nvert"]) * w)
                    assert got.tobytes() == ref[k].tobytes(), ("bit-exact check failed (secondary region, render layouts)", lane, i, k)
        out_bytes_r = int(sum(ca.probe(x).nvert * (12 + 6 + 4 + 8) + ca.probe(x).nface * 6 for x in blobs))
        ms_r = rep_s3.elapsed_s / s_steps * 1e3
        secondary_render = {"mtri_per_s": round(rep_s3.triangles / rep_s3.elapsed_s / 1e6, 2), "mverts_per_s": round(rep_s3.vertices / rep_s3.elapsed_s / 1e6, 2), "ms_per_step": round(ms_r, 4),
                            "steps": s_steps, **window_stats(st_s3, pool_s.lanes), "d2h_bytes_per_step": out_bytes_r, "d2h_GBps": round(out_bytes_r / (ms_r * 1e-3) / 1e9, 2),
                            "d2h_measured_ceiling_GBps": pcie_caps["d2h_GBps"], "pcie_frac": round(out_bytes_r / (ms_r * 1e-3) / 1e9 / pcie_caps["d2h_GBps"], 4),
                            "note": "the secondary region with SURVEY 8f3's render layouts: normals as int16 (upstream's INT16 output format), index as uint16 (Decoder::setIndex(uint16_t *)); "
                                    "positions / uv f32, colours rgba8 as before - algorithmic output bytes of a step %.1f MB instead of %.1f; checked against the oracle's int16 / uint16 outputs" % (out_bytes_r / 1e6, out_bytes_step / 1e6)}
        pool_s.close()
    first_iter = first_iteration() if (rank == 0 and not args.no_other_configs) else None

    # SURVEY 8e's scaling report when N > 1: per-GPU rate, what ONE of the GPUs does alone on the same box right now (same pool shape, the
    # other GPUs idle: every rank but 0 waits at the barrier), efficiency = value / (N x that), and the host-side cost per step
    scaling_block = None
    tri_step = float(sum(ca.probe(x).nface for x in items[0]))
    if world > 1:
        import torch as _t
        mine = _t.zeros(world, dtype=_t.float64, device=red_dev); mine[rank] = float(rep.triangles) / max(rep.elapsed_s, 1e-9) / 1e6
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_gpu = [round(float(x), 2) for x in mine.tolist()]
    else:
        per_gpu = [round(c * tri_step / max(rep.elapsed_s, 1e-9) / 1e6, 2) for c in list(rep.steps_per_device)[:nloc]]
    if n_gpus > 1:
        alone = None
        barrier()
        if rank == 0:
            pool_1 = ca.Pool(devices[:1], threads=nthreads, depth=depth)
            pool_1.set_packed_host_blobs(True)
            pool_1.run(host_items[:1], steps=4 * pool_1.lanes, warmup=0, arenas=None)
            rep_1, _ = pool_1.run(host_items[:1], steps=max(600, 2 * args.steps), warmup=2 * pool_1.lanes, arenas=None)
            alone = rep_1.triangles / rep_1.elapsed_s / 1e6
            pool_1.close()
        barrier()
        scaling_block = {"per_gpu": per_gpu, "alone": alone}                       # (shard.scaling_report, once `value` is known)
    tris_total = shard.sum_over_ranks(float(rep.triangles), dist, red_dev) / R     # (per K-step region)
    verts_total = shard.sum_over_ranks(float(rep.vertices), dist, red_dev) / R
    tris_h = shard.sum_over_ranks(float(rep_h.triangles), dist, red_dev)
    if rank == 0:
        ntri, nvert = int(stats0.total_nface), int(stats0.total_nvert)
        ms_step = elapsed / args.steps * 1e3
        kernels = {k: {"ms_per_step": round(v[0] / solo_steps, 4), "launches_per_step": v[1] // solo_steps} for k, v in kt_acc.items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        # algorithmic bytes of the dominant kernel per launch
        topo_bytes = int(stats0.clers_symbols + stats0.split_bytes + ntri * 12 + nvert * 12)
        alg = {
            # CLERS symbols + split words read; index (12 B/tri) + prediction triples (12 B/vert) written
            "topology_lds": topo_bytes, "topology": topo_bytes,
            "tunstall_decode": int(stats0.tunstall_in + stats0.tunstall_out), "tunstall_tables": int(stats0.tunstall_tables + stats0.tunstall_streams * 9216),
            "tunstall_stream": int(stats0.tunstall_tables + stats0.tunstall_in + stats0.tunstall_out),
        }
        whole_path_bytes = int(stats0.arena_bytes + stats0.output_bytes)
        dom_bytes = alg.get(dom) or whole_path_bytes
        dom_ms = kernels[dom]["ms_per_step"] / max(kernels[dom]["launches_per_step"], 1)
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_note = pmc_traffic(dom)
        out = {
            "metric": "Mtriangles/s + Mverts/s decode, 1M-tri batch; bit-exact vs CPU",
            "value": round(tris_total / elapsed / 1e6, 2), "unit": "Mtri/s",
            "mverts_per_s": round(verts_total / elapsed / 1e6, 2),
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "timed_regions": {"count": R, "ms_per_step": [round(e / args.steps * 1e3, 4) for e in regions],
                              "mean_ms_per_step": round(elapsed_mean / args.steps * 1e3, 4), "median_over_mean": round(elapsed / elapsed_mean, 4),
                              "note": "consecutive regions of exactly K steps each, pipeline full throughout; `value` and `ms_per_step` are the median region's; "
                                      "mean_ms_per_step is all R regions together (a stall anywhere shows there)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/i32 integer + f32 normals",
            "data": "synthetic: 256 distinct bumpy-sphere meshes per GPU (seeds 256*g ..), encoded by the repo's byte-identical .crt writer",
            "config": {"workload": "C4: 256 x (2112 verts / 4096 tris), pos14+uv12+normal10(BORDER)+rgba, per GPU; C5 when n_gpus=8",
                       "blobs_per_gpu": NBLOBS, "tris_per_gpu": ntri, "verts_per_gpu": nvert,
                       "timed_region": "SURVEY 8d primary: K x [H2D of the batch's .crt bytes from ONE pinned host buffer + plan(host walk)+bind+kernels+sync] per GPU, "
                                       "outputs left in HBM; clock from the "
                                       "completion of the last of W warm-up steps to the completion of the K-th timed step, pipeline full at both ends "
                                       "(barrier + device sync before the warm-up and after the drain); K < 100: ~2000/K such regions back to back, the median one reported (timed_regions); "
                                       "the rate with the compressed inputs already resident in HBM (rounds 1-3's `value`) is `resident_inputs`",
                       "h2d_bytes_per_step": int(stats0.arena_bytes), "pcie_GBps": round(stats0.arena_bytes / (ms_step * 1e-3) / 1e9, 2),
                       "pipeline_depth": depth, "host_threads": nthreads, "host_threads_note": threads_note or None, "launch": mode,
                       "parallelism": "blob-sharded x%d, no collective; %s; %d native host threads x %d batches in flight per GPU" % (
                           n_gpus, "one process, one work queue over all GPUs" if world == 1 else "one process per GPU, RCCL only for barrier/max", nthreads, depth)},
            "bit_exact": True, "bit_exact_blobs_checked": checked, "topology_fallbacks": int(rep.topology_fallbacks),
            "steps_per_device": steps_per_device, "per_gpu_mtri_per_s": per_gpu, "numa_local_input_buffers": numa_local_inputs,
            "scaling_report": None,
            "steady_state": window_stats(stamps, pool_lanes),
            "resident_inputs": {"mtri_per_s": round(tris_res / elapsed_res / 1e6, 2), "mverts_per_s": round(tris_res / elapsed_res / 1e6 * nvert / ntri, 2),
                                "ms_per_step": round(elapsed_res / args.steps * 1e3, 4), "regions_ms_per_step": [round(e / args.steps * 1e3, 4) for e in regions_res],
                                "whole_path_GBps": round(whole_path_bytes / (elapsed_res / args.steps) / 1e9, 2),
                                "whole_path_frac_of_8TBps": round(whole_path_bytes / (elapsed_res / args.steps) / 1e9 / 8000.0, 6),
                                **window_stats(stamps_res, pool_lanes),
                                "note": "the same pool and steps with the compressed arena already resident in HBM (no PCIe inside the step): what rounds 1-3 reported as `value`"},
            "tunstall_dictionaries": {"streams": int(stats0.tunstall_streams), "built": int(stats0.tunstall_dictionaries),
                                      "note": "per batch (rebuilt every step): streams of a batch with the same probability table share one dictionary; "
                                              "one generator with 256 seeds repeats tables more than unrelated meshes would - see without_dictionary_sharing"},
            "without_dictionary_sharing": {"mtri_per_s": round(tris_ns / elapsed_ns / 1e6, 2), "ms_per_step": round(elapsed_ns / (fh_steps / nloc) * 1e3, 4), "steps": fh_steps // nloc,
                                           "note": "same pipelined steps and timed region as `value` with $CORTO_TUN_SHARE=2: a dictionary is built for EVERY stream, whatever tables repeat (2 304 per batch instead of ~255) - what a batch of unrelated meshes costs"},
            "irregular_connectivity": irregular,
            "realistic": realistic,
            "secondary_region": secondary,
            "secondary_region_render_layouts": secondary_render,
            "first_iteration": first_iter, "first_iteration_ms": (first_iter or {}).get("first_iteration_ms"),
            "pcie": ({"bytes_per_step": int(stats0.arena_bytes), "GBps": round(stats0.arena_bytes / (ms_step * 1e-3) / 1e9, 2), "measured_ceiling_GBps": pcie_caps["h2d_GBps"],
                      "frac": round(stats0.arena_bytes / (ms_step * 1e-3) / 1e9 / pcie_caps["h2d_GBps"], 4), "ceilings": pcie_caps,
                      "descriptor_bytes_per_step": int(stats0.descriptor_bytes),
                      "GBps_with_descriptors": round((stats0.arena_bytes + stats0.descriptor_bytes) / (ms_step * 1e-3) / 1e9, 2),
                      "note": "the H2D copy inside every timed step (one DMA copy of the batch's compressed bytes) against what back-to-back copies of that size reach on this box; the job descriptors "
                              "are a second, smaller copy on the same link (profiles/r06_what_was_measured.txt: sending a third of them moved nothing - a step from the host is the link AND the GPU, both nearly full)"}
                     if pcie_caps else None),
            "sustained": sustained,
            "poisoned_lanes": int(rep.poisoned_lanes), "pool_warning": pool_warning or None,
            "host_us_per_step_per_thread": round(float(rep.host_us_per_step), 1), "numa_pinned_devices": int(rep.pinned_devices),
            "host_us": {"per_step_per_thread": round(float(rep.host_us_per_step), 1), "plan_walk_bind": round(float(rep.host_plan_us), 1),
                        "wait_for_a_context": round(float(rep.host_wait_us), 1), "harvest": round(float(rep.host_finish_us), 1)},
            "scattered_pageable_blobs": {"mtri_per_s": round(tris_h / elapsed_h / 1e6, 2), "ms_per_step": round(elapsed_h / (fhh_steps / nloc) * 1e3, 4), "steps": fhh_steps // nloc,
                                         **window_stats(stamps_h, pool_lanes), "host_us_per_step_per_thread": round(float(rep_h.host_us_per_step), 1),
                                         "note": "the timed region of `value` with the batch's 256 blobs in 256 separate PAGEABLE host arrays: the worker thread gathers them into "
                                                 "its pinned image (%.1f MB of memcpy) before the DMA copy" % (stats0.arena_bytes / 1e6)},
            "single_batch": {"ms": round(solo_ms, 4), "mtri_per_s": round(ntri / solo_ms / 1e3, 2), "steps": solo_steps,
                             "note": "one batch at a time on one context (latency); `kernels` and `roofline` are measured in this phase",
                             "host_us": {"create_walk": round(stats0.host_create_us, 1), "plan": round(stats0.host_plan_us, 1),
                                         "stage": round(stats0.host_stage_us, 1), "launch": round(stats0.host_launch_us, 1)},
                             "from_host_memory": {"ms": round(h2d_ms, 4), "mtri_per_s": round(ntri / h2d_ms / 1e3, 2),
                                                  "note": "same step with the %.1f MB of compressed blobs uploaded over PCIe inside it" % (stats0.arena_bytes / 1e6)},
                             "to_host_memory": {"ms": round(d2h_ms, 4), "mtri_per_s": round(ntri / d2h_ms / 1e3, 2),
                                                "note": "same step plus the %.1f MB of decoded outputs copied to pinned host memory" % (stats0.output_bytes / 1e6)}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 6), "achievable_peak": 6300.0, "frac_of_achievable": round(ach / 6300.0, 6), "traffic": traffic, "traffic_source": traffic_note, "sources_sha256": sources_sha256(),
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                         "per_step": {"achieved": round(dom_bytes / (elapsed_res / args.steps) / 1e9, 3), "frac": round(dom_bytes / (elapsed_res / args.steps) / 1e9 / 8000.0, 6), "unit": "GB/s",
                                      "ms_per_step": round(elapsed_res / args.steps * 1e3, 4),
                                      "primary_region": {"achieved": round(dom_bytes / (ms_step * 1e-3) / 1e9, 3), "frac": round(dom_bytes / (ms_step * 1e-3) / 1e9 / 8000.0, 6), "ms_per_step": round(ms_step, 4)},
                                      "note": "the same kernel's algorithmic bytes per launch over the PIPELINED step time (one launch of it a step, 2-3 launches of different batches resident at once): "
                                              "what the kernel delivers in the steady state; `achieved` / `frac` above are a launch alone (its latency).  per_step: inputs resident in HBM; primary_region: the step of `value` (H2D inside)"}},
            "whole_path": {"bound": "hbm", "what": "whole path on the timed region of `value` (pinned-host .crt -> HBM outputs)", "algorithmic_bytes": whole_path_bytes,
                           "GBps": round(whole_path_bytes / (ms_step * 1e-3) / 1e9, 2), "peak": 8000.0, "frac_of_8TBps": round(whole_path_bytes / (ms_step * 1e-3) / 1e9 / 8000.0, 6)},
            "kernels": kernels,
            "hbm_ceiling": hbm_ceiling(torch),
        }
        if scaling_block:
            out["scaling_report"] = shard.scaling_report(out["value"], n_gpus, scaling_block["per_gpu"], scaling_block["alone"], out["host_us_per_step_per_thread"])
        if n1_only_skipped:
            out["n1_only_legs"] = "skipped (cpu_baseline, tunstall_scaled, other_configs, irregular / realistic: see the N = 1 line; --all-legs runs them)"
        if share:
            out["shared_gpu"] = True
        if not args.no_tunstall_scaled:
            out["tunstall_scaled"] = tunstall_scaled(ctx, ca, z)
        if not args.no_other_configs and not args.no_tunstall_scaled:
            out["other_configs"] = other_configs(ctx, ca)
            out["encoder_stage"] = encoder_stage(ctx, ca)
            out["facade_per_blob"] = facade_per_blob(ca, blobs, devices[0])
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(blobs)
            out["vs_cpu_1core"] = round(out["value"] / n_gpus / out["cpu_baseline"]["value"], 2)
            if "facade_per_blob" in out:
                out["facade_per_blob"]["cpu_reference_us"] = round(4096 / out["cpu_baseline"]["value"], 1)
        contract_error = None
        try:
            shard.check_bench_line(out, n_gpus)                               # the contract the driver parses (corto_amd/shard.py; tests/test_sharding_cpu.py runs it on an N = 8 line)
        except ValueError as e:                                               # a violated check must not discard a multi-minute measurement: the line is printed WITH the
            contract_error = str(e)                                           # violation named in it, and the exit code says so
            out["contract_error"] = contract_error
        print(json.dumps(out), flush=True)
        if contract_error:
            print("bench.py: " + contract_error, file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and locals().get("contract_error"):
        raise SystemExit(3)


if __name__ == "__main__":
    main()
