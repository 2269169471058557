#!/bin/bash
# copy the summaries of tools/gpu_final.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked): tools/collect_profiles.sh <tag>
TAG=${1:-r06}; cd /root/repo || exit 1
cp gpurun_out/prof_$TAG/kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_per_dispatch.json profiles/${TAG}_pmc_per_dispatch.json
cp gpurun_out/prof_tun_$TAG/pmc_per_dispatch.json profiles/${TAG}_tunstall_scaled_pmc_per_dispatch.json
grep "^{" gpurun_out/${TAG}_final_bench.json | tail -1 > profiles/${TAG}_bench.json
grep "^{" gpurun_out/${TAG}_final_bench_driver.json | tail -1 > profiles/${TAG}_bench_driver_form.json
grep -v "^[EW]2026" gpurun_out/${TAG}_final_prof_pipe.log > profiles/${TAG}_pipelined_kernel_trace.txt
python - <<PY
import json
j = json.load(open("profiles/${TAG}_pmc_per_dispatch.json")); print("pmc stamp", j.get("_sources_sha256", "?")[:12] if isinstance(j, dict) else "?")
b = json.loads(open("profiles/${TAG}_bench_driver_form.json").read()); print("bench stamp", b["roofline"].get("sources_sha256", "?")[:12], "value", b["value"])
PY
