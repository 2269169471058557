import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import corto_amd as ca, bench
blobs, _z = bench.load_blobs(0)
ctx = ca.Context(0)
arena = ca.upload_arena(blobs, 0)
b = ca.Batch(ctx, blobs, device_arena=arena); b.allocate_outputs()
b.decode(); b.sync()
b.decode(); b.sync()
print("descriptor_bytes", b.stats().descriptor_bytes, "arena", b.stats().arena_bytes)
