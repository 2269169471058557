#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04g
timeout 900 python -m pytest tests -m gpu -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -4 ${O}_suite.txt
timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc $?"; tail -3 ${O}_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04g_bench.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "timed_regions", "resident_inputs", "scattered_pageable_blobs", "host_us", "sustained", "irregular_connectivity", "without_dictionary_sharing", "whole_path", "single_batch", "facade_per_blob", "cpu_baseline", "vs_cpu_1core"):
    print(k, json.dumps(j.get(k))[:500])
print("realistic", json.dumps(j.get("realistic"))[:300])
PY
bash tools/prof_run.sh r04 > ${O}_prof_run.log 2>&1; tail -25 ${O}_prof_run.log | cut -c1-250
bash tools/prof_tun.sh r04 > ${O}_prof_tun.log 2>&1; tail -12 ${O}_prof_tun.log | cut -c1-300
bash tools/prof_pipe.sh > ${O}_prof_pipe.log 2>&1; cat ${O}_prof_pipe.log | cut -c1-250
