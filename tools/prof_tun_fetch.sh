export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 python tools/tun_scaled.py 2>&1 | grep -v amdgpu.ids | tail -1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tun_nt; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/p3 -o p3 -- python tools/tun_scaled.py > $OUT/l3.log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/p3/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in acc.items():
        if "tun" in k: print(k, len(n[k]), round(v/len(n[k]),1))
PY
