#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04c
timeout 900 python -m pytest tests -m gpu -x -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -6 ${O}_suite.txt
( timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned-prefetch,pinned,resident,pinned-prefetch
  timeout 200 python tools/fromhost_ab.py 3000 5 4 pinned-prefetch,pinned
  timeout 200 python tools/fromhost_ab.py 3000 4 5 pinned-prefetch,pinned
  timeout 200 python tools/fromhost_ab.py 3000 4 4 scattered-prefetch,scattered ) > ${O}_fromhost_ab.txt 2>&1
grep -v amdgpu.ids ${O}_fromhost_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc $?"; tail -3 ${O}_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04c_bench.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timed_regions", "resident_inputs", "scattered_pageable_blobs", "host_us", "sustained", "irregular_connectivity", "without_dictionary_sharing", "kernels", "tunstall_scaled"):
    print(k, json.dumps(j.get(k))[:700])
print("realistic", json.dumps(j.get("realistic"))[:400])
print("other", json.dumps(j.get("other_configs"))[:900])
PY
