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
