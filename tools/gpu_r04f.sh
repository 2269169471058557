#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04f
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -4 ${O}_suite.txt
( for v in "" corto_amd/lib_uw1/libcorto_hip.so "" corto_amd/lib_uw1/libcorto_hip.so; do echo "LIB=$v"; CORTO_HIP_LIB_PATH=$v GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 5 4 resident,pinned; done
  GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 4 5 resident,pinned
  GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 10 2 resident,pinned
  GPU_MAX_HW_QUEUES=18 timeout 200 python tools/fromhost_ab.py 3000 6 3 resident,pinned
  GPU_MAX_HW_QUEUES=22 timeout 200 python tools/fromhost_ab.py 3000 11 2 resident,pinned
  GPU_MAX_HW_QUEUES=24 timeout 200 python tools/fromhost_ab.py 3000 6 4 resident,pinned
  GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 4 4 resident,pinned ) > ${O}_shapes.txt 2>&1
grep -v "amdgpu.ids\|corto_hip pool" ${O}_shapes.txt | cut -c1-150
