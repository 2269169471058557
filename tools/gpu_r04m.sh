#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04m
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -6 ${O}_suite.txt
python tools/kt_probe.py 2>/dev/null | tail -1
CORTO_TUN_SHARE=2 python tools/kt_probe.py 2>/dev/null | tail -1
bash tools/prof_kernel.sh tun_tables 2>&1 | grep "tun_tables" | cut -c1-300
GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 5 4 resident,pinned 2>&1 | grep -v "amdgpu\|corto_hip pool" | cut -c1-150
CORTO_TUN_SHARE=2 GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 5 4 resident,pinned 2>&1 | grep -v "amdgpu\|corto_hip pool" | cut -c1-150
