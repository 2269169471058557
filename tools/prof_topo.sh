#!/bin/bash
# PMC of the topology kernel alone (unpipelined bench, few steps)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_topo
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT/p -o p -- python bench.py --steps 5 --warmup 1 --no-cpu --no-tunstall-scaled --depth 1 --host-threads 1 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "topology" in k:
            d = {c: x/len(n[k]) for c, x in v.items()}
            sym = 256*4318.0
            print(k, len(n[k]), {c: round(x/sym, 2) for c, x in d.items()})
PY
