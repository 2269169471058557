"""Blob sharding across GPUs: blobs are independent, so the batch is cut into contiguous
ranges balanced by work (nface + nvert, known from the headers); one process per GPU decodes its range;
there is no data-path collective.  torch.distributed is used only for the timing barrier and the
max-over-ranks reduction that bench.py's contract asks for (backend nccl = RCCL on GPUs, gloo in CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def balanced_ranges(weights: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [begin, end) per rank with prefix-sum cut points closest to k/world of the total weight."""
    n = len(weights)
    total = float(sum(weights))
    if world <= 0:
        raise ValueError("world must be positive")
    cuts, acc, k = [0], 0.0, 1
    for i, w in enumerate(weights):
        # put the cut before item i if that is closer to the ideal than after it
        while k < world and acc + w / 2.0 >= k * total / world:
            cuts.append(i)
            k += 1
        acc += w
    while len(cuts) < world:
        cuts.append(n)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def my_range(weights: Sequence[int], world: int, rank: int) -> Tuple[int, int]:
    return balanced_ranges(weights, world)[rank]


def barrier(dist=None, device_sync=None):
    """barrier + device sync on both sides of a timed region"""
    if device_sync is not None:
        device_sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if device_sync is not None:
        device_sync()


def max_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def plan_host_threads(requested: int, gpu_cpus: Sequence[Sequence[int]], affinity: Sequence[int], reserve: int = 1):
    """How many host threads each GPU's decode pool gets (bench.py --host-threads, crthip_pool): every GPU's threads run on the CPUs of ITS
    NUMA node (pool.cpp pins them), so the GPUs of one node share that node's cores.  gpu_cpus[d] = the CPUs next to GPU d ([] = unknown: any
    CPU of `affinity`), affinity = the CPUs this process may run on (os.sched_getaffinity).  Returns (threads per GPU, note).  When some GPU cannot
    even get one core the plan is one time-shared thread each and the note begins with OVERSUBSCRIBED HOST (bench.py prints it to stderr and carries
    it in the JSON line): a scaling run that SILENTLY time-shares its feeder threads would measure the host, not the GPUs."""
    aff = set(int(c) for c in affinity)
    if requested < 1:
        raise ValueError("host threads per GPU must be positive")
    groups = {}
    for d, cpus in enumerate(gpu_cpus):
        usable = tuple(sorted(set(int(c) for c in cpus) & aff)) if cpus else tuple(sorted(aff))
        if not usable:                                    # the node's CPUs are outside this process' cpuset: its threads stay unpinned on what there is
            usable = tuple(sorted(aff))
        groups.setdefault(usable, []).append(d)
    per_gpu, notes = [0] * len(gpu_cpus), []
    for cpus, gpus in groups.items():
        if not cpus:
            raise ValueError("no usable host CPU for GPUs %s" % gpus)
        room = max(len(cpus) - reserve, 0) // len(gpus)
        oversubscribed = room < 1                         # (per GROUP: an oversubscribed node must not hide a later node's "N threads instead of M" note)
        if room < 1:                                      # fewer cores than GPUs: one thread each, time-shared - measured, but said LOUDLY (stderr and the JSON line)
            notes.append("OVERSUBSCRIBED HOST: %d GPU(s) share %d usable host CPU(s) (%s...) - one feeder thread each, time-shared: the rate below measures the host; "
                         "widen the cpuset (taskset / cgroup)" % (len(gpus), len(cpus), ",".join(map(str, cpus[:8]))))
            room = 1
        t = min(requested, room)
        if t < requested and not oversubscribed:
            notes.append("GPUs %s: %d host threads each instead of %d (%d usable CPUs next to them)" % (gpus, t, requested, len(cpus)))
        for d in gpus:
            per_gpu[d] = t
    return per_gpu, "; ".join(notes)


def scaling_report(value_mtri: float, n_gpus: int, per_gpu_mtri, one_gpu_alone_mtri, host_us_per_step_per_thread=None) -> dict:
    """SURVEY 8e's scaling report of an N > 1 bench line: per-GPU rates, what one of the GPUs does alone on the same box with the same pool shape,
    efficiency = value / (N x that).  The driver computes its own efficiency from the per-N lines; this is the in-run figure beside them."""
    if n_gpus < 1 or len(per_gpu_mtri) != n_gpus:
        raise ValueError("per-GPU rates for %d GPUs, got %d" % (n_gpus, len(per_gpu_mtri)))
    rep = {"per_gpu_mtri_per_s": [round(float(x), 2) for x in per_gpu_mtri],
           "one_gpu_alone_mtri_per_s": round(float(one_gpu_alone_mtri), 2) if one_gpu_alone_mtri else None,
           "slowest_over_fastest_gpu": round(min(per_gpu_mtri) / max(per_gpu_mtri), 4) if max(per_gpu_mtri) > 0 else None,
           "note": "per-GPU rate of the timed steps; one_gpu_alone: GPU 0 with the same pool shape and timed region while the other GPUs idle "
                   "(measured behind the main run, the main pool closed); efficiency_vs_1gpu = value / (n_gpus x one_gpu_alone)"}
    if one_gpu_alone_mtri:
        rep["efficiency_vs_1gpu"] = round(float(value_mtri) / (n_gpus * float(one_gpu_alone_mtri)), 4)
    if host_us_per_step_per_thread is not None:
        rep["host_us_per_step_per_thread"] = round(float(host_us_per_step_per_thread), 1)
    return rep


BENCH_LINE_KEYS = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": (int, float),
                   "higher_is_better": bool, "scaling": str, "vs_baseline": (type(None), int, float), "dtype": str, "data": str, "config": dict}


def check_bench_line(line: dict, n_gpus: int = None) -> None:
    """the contract every bench.py line keeps (the driver parses it): the named keys with their types, config.workload and no model keys,
    roofline with achieved / peak / frac consistent, scaling "weak", and at N > 1 a scaling report with one rate per GPU.  Raises ValueError."""
    for k, t in BENCH_LINE_KEYS.items():
        if k not in line:
            raise ValueError("bench line: key %r missing" % k)
        if not isinstance(line[k], t) or (t is int and isinstance(line[k], bool)):
            raise ValueError("bench line: %r is %r" % (k, type(line[k]).__name__))
    if "workload" not in line["config"] or "model" in line["config"]:
        raise ValueError("bench line: config needs `workload` and no model keys")
    if line["scaling"] != "weak" or line["higher_is_better"] is not True or line["unit"] != "Mtri/s":
        raise ValueError("bench line: scaling / higher_is_better / unit")
    if n_gpus is not None and line["n_gpus"] != n_gpus:
        raise ValueError("bench line: n_gpus %r, expected %r" % (line["n_gpus"], n_gpus))
    if not (line["value"] > 0 and line["ms_per_step"] > 0):
        raise ValueError("bench line: value / ms_per_step not positive")
    r = line.get("roofline")
    if not isinstance(r, dict) or r.get("bound") not in ("hbm", "mfma") or r.get("unit") not in ("GB/s", "TFLOP/s"):
        raise ValueError("bench line: roofline block")
    if abs(r["achieved"] / r["peak"] - r["frac"]) > 1e-4 or not ("traffic" in r):
        raise ValueError("bench line: roofline.frac is not achieved / peak")
    if line["n_gpus"] == 1:
        cb = line.get("cpu_baseline")
        if cb is not None and not (isinstance(cb, dict) and {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] in ("reference", "port")):
            raise ValueError("bench line: cpu_baseline block")
    else:
        if "cpu_baseline" in line:
            raise ValueError("bench line: cpu_baseline belongs to the N = 1 line")
        sr = line.get("scaling_report")
        if not isinstance(sr, dict) or len(sr.get("per_gpu_mtri_per_s", [])) != line["n_gpus"]:
            raise ValueError("bench line: scaling_report needs one rate per GPU")
        if not all(x > 0 for x in sr["per_gpu_mtri_per_s"]):                  # (their sum is over ALL timed regions, `value` the median region: not compared)
            raise ValueError("bench line: a GPU decoded nothing")
