#!/bin/bash
# rocprofv3 kernel trace of the scaled Tunstall run (tools/tun_scaled.py), single pass and ($CORTO_TUN_TWO_PASS=1) two passes
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_tun_trace
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for tp in 0 1; do
rm -rf $OUT/t$tp; CORTO_TUN_TWO_PASS=$tp rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/t$tp -o t -- python tools/tun_scaled.py > $OUT/l$tp.log 2>&1
echo "two_pass=$tp"; python - <<PY
import csv, glob
f = glob.glob("$OUT/t$tp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tun" in r["Name"] or "scan" in r["Name"]: print("  %-60s calls %3s avg %8.1f us min %8.1f" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
tail -1 $OUT/l$tp.log | cut -c1-200
done
