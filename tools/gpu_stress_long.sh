#!/bin/bash
# a longer random-mesh run than gpu_stress.sh (several seeds, u16 / u32, both passes): after a change to the automaton's wave-wide steps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for sd in 11 12 13 14 15 16; do timeout 900 python tools/stress_topology.py 16 $sd 2>&1 | grep -v "persistent\|amdgpu" | tail -3; done > gpurun_out/stress_long.txt 2>&1
cat gpurun_out/stress_long.txt
