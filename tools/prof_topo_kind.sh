#!/bin/bash
# PMC of the topology kernel on a batch of one kind of mesh (tools/delta_probe.py): usage prof_topo_kind.sh "holey disc" <symbols per blob>
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_topo_kind
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export KIND="$1"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT/p -o p -- python tools/delta_probe.py > $OUT/log.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH -d $OUT/p -o q -- python tools/delta_probe.py > $OUT/log2.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT -d $OUT/p -o r -- python tools/delta_probe.py > $OUT/log3.txt 2>&1
python - <<PY
import csv, glob, collections
sym = 256.0*float("${2:-4318}")
for f in sorted(glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    # the LAST dispatch only: the first decodes of a context may still redo blobs on the HBM front (the LDS slots adapt)
    rows = [r for r in csv.DictReader(open(f)) if "topology" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    d = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) == last: d[r["Counter_Name"]] += float(r["Counter_Value"])
    print("last dispatch, per symbol:", {c: round(x/sym, 2) for c, x in d.items()})
PY
tail -3 $OUT/log.txt
