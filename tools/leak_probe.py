#!/usr/bin/env python3
"""Host RSS against steps: the pool with resident inputs, from one pinned buffer, and a bare loop of one batch object decoded again and again on one context.
   python tools/leak_probe.py [steps per slice] [slices]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import corto_amd as ca
def rss_mb():
    with open("/proc/self/statm") as f: return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20
per = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
slices = int(sys.argv[2]) if len(sys.argv) > 2 else 4
blobs, _ = bench.load_blobs(0)
pin, views = ca.pinned_host_arena(blobs)
arena = [[ca.upload_arena(blobs, 0)]]
for name, items, arenas, packed in (("resident", blobs, arena, False), ("pinned", views, None, True)):
    pool = ca.Pool([0], threads=5, depth=4)
    pool.set_packed_host_blobs(packed)
    pool.run([items], steps=pool.lanes * 8, warmup=0, arenas=arenas)
    r0 = rss_mb(); out = []
    for _ in range(slices):
        rep, st = pool.run([items], steps=per, warmup=0, arenas=arenas); del st
        out.append(round(rss_mb() - r0, 1))
    print("pool %-9s RSS growth (MiB) after each %d steps: %s -> %.0f bytes a step" % (name, per, out, out[-1] * 2**20 / (per * slices)), flush=True)
    pool.close()
ctx = ca.Context(0)
b = ca.Batch(ctx, blobs, device_arena=arena[0][0]); b.allocate_outputs()
for _ in range(50): b.decode(); b.sync()
r0 = rss_mb(); out = []
n = max(1000, per // 20)
for _ in range(slices):
    for _ in range(n): b.decode(); b.sync()
    out.append(round(rss_mb() - r0, 1))
print("one batch object, decode + sync, RSS growth after each %d decodes: %s -> %.0f bytes a decode" % (n, out, out[-1] * 2**20 / (n * slices)))
import torch
x = torch.empty(1 << 20, dtype=torch.uint8).pin_memory(); y = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
for _ in range(100): y.copy_(x, non_blocking=True); y.add_(1)
torch.cuda.synchronize(); r0 = rss_mb(); out = []
for _ in range(slices):
    for _ in range(n):
        y.copy_(x, non_blocking=True); y.add_(1); torch.cuda.synchronize()
    out.append(round(rss_mb() - r0, 1))
print("torch: pinned copy + kernel + sync, RSS growth after each %d: %s -> %.0f bytes an iteration" % (n, out, out[-1] * 2**20 / (n * slices)))
