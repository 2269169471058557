#!/bin/bash
# PMC of every kernel of the decode on a batch of 256 blobs of ONE kind of mesh (tools/kt_probe_irregular.py: $MESH = flipped | torus | holey | strip |
# grid128, $FLIP): instructions per dispatch, the last dispatch of each kernel (the first decodes of a context may still redo blobs on the HBM front).
# usage: MESH=flipped FLIP=0.5 bash tools/prof_topo_kind.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_topo_kind
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/p -o p -- python tools/kt_probe_irregular.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    last = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]; last[k] = max(last.get(k, 0), int(r["Dispatch_Id"]))
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Dispatch_Id"]) == last[k]: d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in d.items():
        if "corto_hip" in k: print(k.replace("corto_hip::", ""), {c: round(x) for c, x in v.items()})
PY
grep -v amdgpu $OUT/log.txt | tail -2 | cut -c1-300
rm -rf $OUT/p
