"""Stand-alone scaled Tunstall run (the tunstall_scaled leg of bench.py) for profiling."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import corto_amd as ca
import bench
ctx = ca.Context(0)
ctx.set_profiling(True)
print(json.dumps(bench.tunstall_scaled(ctx, ca, None)))
