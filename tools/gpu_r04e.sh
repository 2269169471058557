#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04e
timeout 900 python -m pytest tests -m gpu -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -8 ${O}_suite.txt
( timeout 200 python tools/fromhost_ab.py 3000 5 4 pinned,resident
  GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 5 4 pinned,resident
  timeout 200 python tools/fromhost_ab.py 3000 6 4 pinned
  timeout 200 python tools/fromhost_ab.py 3000 5 5 pinned
  timeout 200 python tools/fromhost_ab.py 3000 7 3 pinned ) > ${O}_fromhost_ab.txt 2>&1
grep -v amdgpu.ids ${O}_fromhost_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc $?"; tail -3 ${O}_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04e_bench.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timed_regions", "resident_inputs", "scattered_pageable_blobs", "host_us", "sustained", "irregular_connectivity", "without_dictionary_sharing", "whole_path"):
    print(k, json.dumps(j.get(k))[:600])
print("realistic", json.dumps(j.get("realistic"))[:400])
PY
