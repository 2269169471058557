#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_run.sh <tag>
# writes rocprofv3 summaries under gpurun_out/prof_<tag>; the ones to be judged are copied into profiles/.
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu --no-other-configs --sustain 0 --depth 1 --host-threads 1"   # unpipelined: every kernel runs alone, as in bench.py's `kernels` phase
timeout 300 timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.log 2>&1
# PMC passes: counters only (no trace domains), one small group per pass
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- $BENCH --no-tunstall-scaled > $OUT/bench_pmc1.log 2>&1
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $BENCH --steps 5 > $OUT/bench_fetch.log 2>&1
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $BENCH --steps 5 > $OUT/bench_write.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
for tag in ("pmc1", "pmc_fetch", "pmc_write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: x / len(n[k]) for c, x in v.items()}); out[k]["dispatches_" + tag] = len(n[k])
import sys; sys.path.insert(0, "."); import bench
out["_sources_sha256"] = bench.sources_sha256()          # ties the counters to the kernels they were taken from (bench.py refuses a mismatch)
json.dump(out, open("$OUT/pmc_per_dispatch.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict): print(k, {c: round(x, 1) for c, x in v.items()})
PY
tail -1 $OUT/bench_trace.log | cut -c1-200
# keep the summaries, drop the per-dispatch CSVs (tens of MB: gpurun copies back 64 MiB at most)
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc_fetch $OUT/pmc_write
