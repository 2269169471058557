#!/bin/bash
# how long the untimed pre-warm has to be for a 20-step timed region to see the steady-state rate (bench.py: BENCH_PREWARM_STEPS)
cd $GRAFT_REPO_ROOT
for pw in ${PW:-128 128 128 4000 128}; do
  echo -n "prewarm $pw: "
  BENCH_PREWARM_STEPS=$pw python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu --no-other-configs --no-tunstall-scaled --sustain 0 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(j['value'], j['timed_regions']['ms_per_step'])"
done
