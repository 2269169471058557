#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04b
timeout 120 python tools/h2d_probe.py > ${O}_h2d.txt 2>&1; cat ${O}_h2d.txt | grep H2D
HSA_ENABLE_SDMA=0 timeout 120 python tools/h2d_probe.py > ${O}_h2d_blit.txt 2>&1; echo "--- HSA_ENABLE_SDMA=0"; cat ${O}_h2d_blit.txt | grep H2D
( timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned-prefetch,pinned
  GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned-prefetch,pinned,resident
  GPU_MAX_HW_QUEUES=24 timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned-prefetch,pinned,resident
  timeout 200 python tools/fromhost_ab.py 3000 4 3 pinned-prefetch,pinned,resident
  timeout 200 python tools/fromhost_ab.py 3000 6 3 pinned
  timeout 200 python tools/fromhost_ab.py 3000 5 4 pinned
  HSA_ENABLE_SDMA=0 timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned-prefetch,pinned ) > ${O}_fromhost_ab.txt 2>&1
grep -v amdgpu.ids ${O}_fromhost_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -4 ${O}_suite.txt
timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc $?"; tail -3 ${O}_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04b_bench.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timed_regions", "resident_inputs", "scattered_pageable_blobs", "host_us", "sustained", "irregular_connectivity", "without_dictionary_sharing", "kernels", "roofline", "facade_per_blob", "single_batch"):
    print(k, json.dumps(j.get(k))[:700])
print("realistic", json.dumps(j.get("realistic"))[:400])
PY
