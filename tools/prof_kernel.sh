#!/bin/bash
# PMC of one kernel (name substring $1) of the unpipelined bench: instruction mix, waits, LDS behaviour
export TMPDIR=/tmp
K=${1:-normal_blob}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_kernel
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-tunstall-scaled --no-other-configs --depth 1 --host-threads 1"
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d $OUT/p -o p -- $B > $OUT/log1.txt 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $OUT/p -o q -- $B > $OUT/log2.txt 2>&1
timeout 300 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL -d $OUT/p -o r -- $B > $OUT/log3.txt 2>&1
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "$K" in k: print(k, len(n[k]), {c: round(x/len(n[k])) for c, x in v.items()})
PY
rm -rf $OUT/p
