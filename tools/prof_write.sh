#!/bin/bash
# WRITE_SIZE / FETCH_SIZE per dispatch of every kernel of the unpipelined bench (two --pmc passes of their own): what a change does to a kernel's HBM traffic
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_write
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-tunstall-scaled --no-other-configs --depth 1 --host-threads 1"
timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/p -o w -- $B > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/p -o f -- $B > $OUT/log2.txt 2>&1
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "corto_hip" in k: print(k, len(n[k]), {c: round(x/len(n[k])) for c, x in v.items()})
PY
rm -rf $OUT/p
