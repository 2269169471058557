#!/bin/bash
# kernel trace of ONE batch at a time on a lone (two-stream) context: where a single batch's latency goes - kernels on the critical path and the gaps between them
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_single
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rm -rf $OUT/t; timeout 300 rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o t -- python tools/kt_probe.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "corto_hip" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# batches: split at k_tun_tables launches that follow a k_normal_blob
batches, cur = [], []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("corto_hip::", "")
    cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if n == "k_normal_blob": batches.append(cur); cur = []
for b in batches[-3:]:
    t0 = b[0][1]
    print("batch: span %.1f us, kernels' own time %.1f us" % ((b[-1][2] - t0) / 1e3, sum(e - s for _, s, e in b) / 1e3))
    for n, s, e in b: print("   %-26s start %7.1f  end %7.1f  (%.1f us)" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
tail -1 $OUT/log.txt | cut -c1-300
rm -rf $OUT/t
