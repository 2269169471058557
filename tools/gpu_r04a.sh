#!/bin/bash
# round 4, first GPU call: suite + driver-form bench + the from-host A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_suite.txt 2>&1; echo "suite rc $?" >> gpurun_out/r04a_suite.txt
tail -5 gpurun_out/r04a_suite.txt
timeout 300 python tools/fromhost_ab.py > gpurun_out/r04a_fromhost_ab.txt 2>&1; tail -20 gpurun_out/r04a_fromhost_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r04a_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04a_bench.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timed_regions", "resident_inputs", "scattered_pageable_blobs", "host_us", "sustained", "irregular_connectivity", "without_dictionary_sharing", "kernels", "roofline", "facade_per_blob"):
    print(k, json.dumps(j.get(k))[:600])
print("realistic", json.dumps(j.get("realistic"))[:400])
PY
