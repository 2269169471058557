#!/bin/bash
# PMC of the topology kernel alone (unpipelined bench, few steps)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_topo
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-tunstall-scaled --no-other-configs --depth 1 --host-threads 1"
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT/p -o p -- $B > $OUT/log.txt 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -d $OUT/p -o q -- $B > $OUT/log2.txt 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS -d $OUT/p -o r -- $B > $OUT/log3.txt 2>&1
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
