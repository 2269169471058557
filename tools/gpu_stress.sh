#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 500 python tools/stress_topology.py 14 4 2>&1 | tail -4
  CORTO_DELTA_WIDE=1 timeout 400 python tools/stress_topology.py 8 5 2>&1 | tail -3
  CORTO_UNPACK_CHUNKED=1 timeout 300 python tools/stress_topology.py 5 6 2>&1 | tail -3
  CORTO_DELTA_ROUNDS=1 timeout 400 python tools/stress_topology.py 8 7 2>&1 | tail -3
  CORTO_DELTA_ROUNDS=1 CORTO_DELTA_WIDE=1 timeout 300 python tools/stress_topology.py 5 8 2>&1 | tail -3
  for sd in 1 2 3 4 5 6; do SEED=$sd NMUT=96 timeout 120 python tools/fuzz_probe.py 2>&1 | tail -2; CORTO_DELTA_WIDE=1 SEED=1$sd NMUT=96 timeout 120 python tools/fuzz_probe.py 2>&1 | tail -1; done ) > gpurun_out/r06_stress.txt 2>&1
grep -v amdgpu gpurun_out/r06_stress.txt
