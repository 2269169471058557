#!/bin/bash
# usage: tests/prof_run.sh <tag>   (runs on the GPU box; writes summaries under gpurun_out/prof_<tag>)
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --steps 10 --warmup 2 --no-cpu --no-tunstall-scaled > $OUT/bench_trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tunstall-scaled > $OUT/bench_pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -d $OUT/pmc2 -o pmc2 -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tunstall-scaled > $OUT/bench_pmc2.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for tag in ("pmc1","pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in acc.items():
            if "topology" in k or "delta" in k or "tun_tables" in k:
                print(tag, k, dict(v))
PY
