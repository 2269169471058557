#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pipe
mkdir -p $OUT; cd $GRAFT_REPO_ROOT
# the same pool, un-profiled first: how much the profiler perturbs the operating point is part of the evidence
PROBE_LONG=${STEPS:-2000} GPU_MAX_HW_QUEUES=20 python tools/pool_probe.py ${SHAPE:-5 4} 2>/dev/null | tail -1 | sed 's/^/un-profiled: /'
rm -rf $OUT/t; PROBE_LONG=${STEPS:-2000} GPU_MAX_HW_QUEUES=20 timeout 300 rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o t -- python tools/pool_probe.py ${SHAPE:-5 4} > $OUT/log.txt 2>&1
grep "ms/step" $OUT/log.txt | sed 's/^/under rocprofv3: /'
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
topo = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_topology_lds" in r["Kernel_Name"]]
# steady state = the last 40 topology launches
ph = topo[len(topo) // 4: 3 * len(topo) // 4]                      # steady state: the middle half of the long run
t0, t1 = ph[0][0], ph[-1][1]
print("steady state: %d topology launches in %.2f ms -> %.3f ms/step" % (len(ph), (t1 - t0) / 1e6, (t1 - t0) / 1e6 / len(ph)))
ev = sorted([(s, 1) for s, e in ph] + [(e, -1) for s, e in ph])
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
print("time share by number of concurrent topology kernels:", {k: round(v / (t1 - t0), 3) for k, v in sorted(hist.items())})
print("topology duration avg %.0f us" % (sum(e - s for s, e in ph) / len(ph) / 1e3))
allk = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
ev = sorted([(s, 1) for s, e in allk] + [(e, -1) for s, e in allk])
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[min(cur, 12)] += t - last; last = t; cur += d
print("time share by number of concurrent kernels (any):", {k: round(v / (t1 - t0), 3) for k, v in sorted(hist.items())})
acc = collections.defaultdict(list)
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 <= s <= t1: acc[r["Kernel_Name"].split("(")[0].replace("corto_hip::", "")].append((int(r["End_Timestamp"]) - s) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s n=%4d avg %8.1f us max %8.1f" % (k, len(v), sum(v) / len(v), max(v)))
PY
tail -4 $OUT/log.txt | head -4
