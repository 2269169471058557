"""Stand-alone scaled Tunstall run (the tunstall_scaled leg of bench.py) for profiling."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
import bench
ctx = ca.Context(0)
ctx.set_profiling(True)
for m in [int(x) for x in os.environ.get("ABL", "0").split(",")]:
    tid = [int(x) for x in os.environ["TABLES"].split(",")] if os.environ.get("TABLES") else None
    r = bench.tunstall_scaled(ctx, ca, None, tid)
    print(m, r["decode_kernel_ms"], r["decode_kernel_GBps"], r["bytes_written"], "all kernels ms", r["all_tunstall_kernels_ms"], r.get("kernels_ms"))
