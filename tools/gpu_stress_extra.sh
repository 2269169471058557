cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# a wider campaign on the final build: more seeds of every setting (each line: 480 random-mesh decodes against the oracle, byte for byte)
( for sd in $(seq 101 120); do echo -n "default seed $sd: "; timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | grep -v amdgpu | tail -1 | cut -c1-60; done
  for sd in $(seq 121 128); do echo -n "CORTO_DELTA_ROUNDS=1 seed $sd: "; CORTO_DELTA_ROUNDS=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 129 134); do echo -n "CORTO_VALUES_I32=1 seed $sd: "; CORTO_VALUES_I32=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 135 140); do echo -n "CORTO_DELTA_WIDE=1 seed $sd: "; CORTO_DELTA_WIDE=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 141 146); do echo -n "CORTO_UNPACK_CHUNKED=1 seed $sd: "; CORTO_UNPACK_CHUNKED=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 147 152); do echo -n "CORTO_TUN_SHARE=2 seed $sd: "; CORTO_TUN_SHARE=2 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 153 156); do echo -n "CORTO_DELTA_WALK=1 seed $sd: "; CORTO_DELTA_WALK=1 timeout 300 python tools/stress_topology.py 10 $sd 2>&1 | tail -1 | cut -c1-60; done
  for sd in $(seq 31 46); do echo -n "fuzz seed $sd: "; SEED=$sd NMUT=128 timeout 120 python tools/fuzz_probe.py 2>&1 | tail -1; done ) > gpurun_out/r06_stress_extra.txt 2>&1
grep -c "mismatching arrays 0" gpurun_out/r06_stress_extra.txt; grep -v "mismatching arrays 0\|fuzz probe ok" gpurun_out/r06_stress_extra.txt | head
