"""CPU, world_size 2 over gloo: the N>1 protocol bench.py uses - contiguous work-balanced blob ranges, every rank
decodes only its own range (here with the CPU oracle standing in for the device), barrier-bracketed timing,
MAX over ranks for time and SUM for work.  No data-path collective exists to test: blobs share nothing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT
from corto_amd import shard


def test_balanced_ranges_partition_and_balance():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 17, 256, 2048):
            w = rng.integers(1, 100, n).tolist()
            r = shard.balanced_ranges(w, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            if n >= 64:
                loads = [sum(w[a:b]) for a, b in r]
                assert max(loads) - min(loads) <= 2 * max(w)
    assert shard.balanced_ranges([6208] * 2048, 8) == [(k * 256, (k + 1) * 256) for k in range(8)]   # C5 -> 8 x C4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as oc
    from corto_amd import shard as sh
    z = np.load(os.path.join(GOLDEN, "c4_blobs16.npz"))
    from conftest import aligned
    blobs = [aligned(z["crt_%02d" % s]) for s in range(16)]
    weights = [oc.parse_header(b)["nface"] + oc.parse_header(b)["nvert"] for b in blobs]
    a, b = sh.my_range(weights, world, rank)
    import time
    sh.barrier(dist)
    t0 = time.perf_counter()
    tris = 0
    for i in range(a, b):
        tris += oc.decode(blobs[i])["nface"]
    elapsed = time.perf_counter() - t0 + 0.01 * rank            # make the ranks differ
    sh.barrier(dist)
    tmax = sh.max_over_ranks(elapsed, dist)
    total = sh.sum_over_ranks(float(tris), dist)
    out[rank] = (a, b, elapsed, tmax, total)
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (a0, b0, e0, m0, t0), (a1, b1, e1, m1, t1) = out[0], out[1]
    assert (a0, b0, a1, b1) == (0, 8, 8, 16)                     # 16 equal blobs -> two halves, no overlap
    assert m0 == m1 == max(e0, e1)                               # every rank sees the slowest rank's time
    assert t0 == t1 == 16 * 4096                                 # whole-job work is the sum over ranks


def test_host_threads_are_sized_for_the_gpus_that_share_a_socket():
    """bench.py --gpus 8: five feeder threads a GPU are 40 threads on two sockets - fine on a 2 x 128-core host, too many for a cpuset of 16; the
    plan gives every GPU of a NUMA node an equal share of that node's usable CPUs (one kept back), and says OVERSUBSCRIBED when a GPU could not get one"""
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    gpus = [node0] * 4 + [node1] * 4
    t, note = shard.plan_host_threads(5, gpus, range(256))
    assert t == [5] * 8 and note == ""
    t, note = shard.plan_host_threads(5, gpus, list(range(0, 8)) + list(range(64, 72)))       # a cpuset of 8 + 8 CPUs: (8 - 1) // 4 = 1 thread a GPU
    assert t == [1] * 8 and "instead of 5" in note
    t, _ = shard.plan_host_threads(5, gpus[:1], range(256))
    assert t == [5]
    t, _ = shard.plan_host_threads(6, [[]] * 2, range(12))                                     # NUMA unknown: the process' CPUs shared by all GPUs
    assert t == [5, 5]
    t, note = shard.plan_host_threads(5, gpus, range(0, 24))                                   # node 1's CPUs are outside the cpuset: its GPUs' threads run unpinned on the same 24 as node 0's
    assert t == [2] * 8 and "instead of 5" in note
    t, note = shard.plan_host_threads(5, gpus, range(0, 4))                                    # 8 GPUs on 4 CPUs: still runs, one time-shared thread each, and says so
    assert t == [1] * 8 and note.startswith("OVERSUBSCRIBED HOST")
    with pytest.raises(ValueError):
        shard.plan_host_threads(0, gpus, range(256))


def _canned_line(n_gpus):
    """a bench line as bench.py assembles it, from the last driver-run record's numbers"""
    import json
    rec = json.load(open(os.path.join(ROOT, "BENCH_r04.json")))["parsed"]
    line = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")}
    if n_gpus > 1:
        per_gpu = [line["value"] * (0.93 + 0.01 * g) for g in range(n_gpus)]
        line.update(n_gpus=n_gpus, value=round(sum(per_gpu), 2), ms_per_step=line["ms_per_step"] / 0.96)
        del line["cpu_baseline"]
        line["scaling_report"] = shard.scaling_report(line["value"], n_gpus, per_gpu, rec["value"], 190.0)
    return line


def test_bench_line_contract_at_one_and_eight_gpus():
    """what the driver parses: shard.check_bench_line (bench.py runs it on every line it prints) on an N = 1 line and on the N = 8 line assembled from
    eight per-GPU rates - schema, roofline arithmetic, the scaling report's one rate per GPU adding up to `value`, efficiency against the N = 1 rate"""
    one = _canned_line(1)
    shard.check_bench_line(one, 1)
    eight = _canned_line(8)
    shard.check_bench_line(eight, 8)
    sr = eight["scaling_report"]
    assert len(sr["per_gpu_mtri_per_s"]) == 8 and 0.9 < sr["efficiency_vs_1gpu"] < 1.01 and sr["slowest_over_fastest_gpu"] < 1
    assert abs(sr["efficiency_vs_1gpu"] - eight["value"] / (8 * one["value"])) < 1e-3
    for breakage in (lambda l: l.pop("roofline"), lambda l: l.update(scaling="strong"), lambda l: l["config"].update(model="x"), lambda l: l.update(n_gpus=4),
                     lambda l: l["scaling_report"]["per_gpu_mtri_per_s"].pop(), lambda l: l.update(cpu_baseline={}), lambda l: l.update(value=0)):
        import copy
        bad = copy.deepcopy(eight)
        breakage(bad)
        with pytest.raises(ValueError):
            shard.check_bench_line(bad, 8)
    with pytest.raises(ValueError):
        shard.scaling_report(1.0, 8, [1.0] * 7, 1.0)
