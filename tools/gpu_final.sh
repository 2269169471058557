#!/bin/bash
# the round's evidence in one GPU call: GPU suite, rocprofv3 summaries (tools/prof_*.sh <tag>), the driver-form bench, the default bench
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/${TAG}_final
timeout 900 python -m pytest tests -m gpu -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -3 ${O}_suite.txt
bash tools/prof_run.sh $TAG > ${O}_prof_run.log 2>&1; tail -22 ${O}_prof_run.log | cut -c1-260
# the counters of THIS build where bench.py looks for them (roofline.traffic is reported only from a file whose stamp is the running build's)
cp gpurun_out/prof_$TAG/pmc_per_dispatch.json profiles/${TAG}_pmc_per_dispatch.json
timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench_driver.json 2> ${O}_bench_driver.err; echo "bench (driver form) rc $?"
timeout 600 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench (default) rc $?"
python - <<PY
import json
for f in ("${O}_bench_driver.json", "${O}_bench.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f)
    for k in ("value", "mverts_per_s", "ms_per_step", "timed_regions", "resident_inputs", "sustained", "pcie", "irregular_connectivity", "without_dictionary_sharing", "realistic", "secondary_region", "secondary_region_render_layouts", "first_iteration", "scattered_pageable_blobs", "whole_path", "roofline", "single_batch", "kernels", "tunstall_scaled", "other_configs", "facade_per_blob", "cpu_baseline", "vs_cpu_1core", "host_us", "hbm_ceiling"):
        print("  ", k, json.dumps(j.get(k))[:700])
PY
bash tools/prof_tun.sh $TAG > ${O}_prof_tun.log 2>&1; grep "^p[1-4] " ${O}_prof_tun.log | cut -c1-330
bash tools/prof_pipe.sh > ${O}_prof_pipe.log 2>&1; grep -v "^[EW]2026" ${O}_prof_pipe.log | cut -c1-250
rm -rf gpurun_out/prof_pipe/t
