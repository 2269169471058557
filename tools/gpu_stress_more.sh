cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for sd in 31 32 33 34 35 36; do timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | grep -v amdgpu | tail -3; done
  for sd in 41 42 43; do CORTO_DELTA_ROUNDS=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | grep -v amdgpu | tail -3; done
  for sd in 51 52; do CORTO_TUN_SHARE=2 timeout 300 python tools/stress_topology.py 8 $sd 2>&1 | grep -v amdgpu | tail -3; done
  for sd in 21 22 23 24 25 26 27 28; do SEED=$sd NMUT=128 timeout 120 python tools/fuzz_probe.py 2>&1 | tail -1; done ) > gpurun_out/r06_stress_more.txt 2>&1
cat gpurun_out/r06_stress_more.txt
