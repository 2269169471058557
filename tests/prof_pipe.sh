#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pipe
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=16 rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o t -- python tests/pipe_probe.py > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# last 400 kernels = depth 4 phase
tail = rows[-330:]
b = int(tail[0]["Start_Timestamp"])
for r in tail[:110]:
    n = r["Kernel_Name"].split("(")[0].replace("corto_hip::", "")
    print("%-22s q%-3s start %9.1f dur %8.1f us" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - b) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
tail -4 $OUT/log.txt
