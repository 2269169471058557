#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tun_$1
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p1 -o p1 -- python tools/tun_scaled.py > $OUT/l1.log 2>&1
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o p2 -- python tools/tun_scaled.py > $OUT/l2.log 2>&1
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/p3 -o p3 -- python tools/tun_scaled.py > $OUT/l3.log 2>&1
timeout 300 timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/p4 -o p4 -- python tools/tun_scaled.py > $OUT/l4.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {}
for tag in ("p1","p2","p3","p4"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            if "tun" in k:
                print(tag, k, len(n[k]), {c: round(x/len(n[k]),1) for c, x in v.items()})
                out.setdefault(k, {}).update({c: x/len(n[k]) for c, x in v.items()}); out[k]["dispatches_" + tag] = len(n[k])
import sys; sys.path.insert(0, "."); import bench
out["_sources_sha256"] = bench.sources_sha256()          # ties the counters to the kernels they were taken from (bench.py refuses a mismatch)
json.dump(out, open("$OUT/pmc_per_dispatch.json", "w"), indent=1)
PY
tail -1 $OUT/l1.log | cut -c1-300
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
