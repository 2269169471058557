#!/bin/bash
# where a context's idle time between two steps goes: kernel trace + HIP API trace of the pipelined pool (tools/pool_probe.py), joined
# by correlation id: end of a step's last kernel -> the next step's first hipLaunchKernel call (host reaction + planning) -> that
# kernel's start on the GPU (queueing)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gap
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
rm -rf $OUT/t; PROBE_LONG=${STEPS:-1500} GPU_MAX_HW_QUEUES=16 rocprofv3 --output-format csv --kernel-trace --hip-runtime-trace -d $OUT/t -o t -- python tools/pool_probe.py ${SHAPE:-4 4} > $OUT/log.txt 2>&1
grep "ms/step" $OUT/log.txt
ls $OUT/t
python - <<PY
import csv, glob, collections, statistics
kt = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
at = glob.glob("$OUT/t/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
rows = list(csv.DictReader(open(at)))
print("api columns", list(rows[0].keys()))
byfn = collections.defaultdict(list)
for r in rows:
    api[r["Correlation_Id"]] = (r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Thread_Id"))
    byfn[r["Function"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for f, v in sorted(byfn.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-34s n=%6d total %8.1f ms  avg %7.1f us  p90 %7.1f" % (f, len(v), sum(v) / 1e3, sum(v) / len(v), sorted(v)[int(len(v) * 0.9)]))
k = list(csv.DictReader(open(kt)))
byq = collections.defaultdict(list)
for r in k:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("corto_hip::", ""), r["Correlation_Id"]))
react, queue, first_to_last_launch = [], [], []
for q, v in byq.items():
    v.sort()
    if len(v) < 400: continue
    mid = v[len(v) // 4: 3 * len(v) // 4]
    for (s0, e0, n0, c0), (s1, e1, n1, c1) in zip(mid, mid[1:]):
        if n0 == "k_normal_blob" and n1 == "k_tun_tables" and c1 in api:
            f, a0, a1, th = api[c1]
            react.append((a0 - e0) / 1e3); queue.append((s1 - a0) / 1e3)
def st(x): return "med %.0f mean %.0f p10 %.0f p90 %.0f" % (statistics.median(x), statistics.mean(x), sorted(x)[len(x) // 10], sorted(x)[9 * len(x) // 10])
print("last kernel's end -> next step's first launch call (us):", st(react))
print("first launch call -> that kernel's start on the GPU (us):", st(queue))
PY
