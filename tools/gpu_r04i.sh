#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04i
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -4 ${O}_suite.txt
python tools/kt_probe.py 2>/dev/null | tail -1
bash tools/prof_kernel.sh ${1:-delta_lds16} 2>&1 | grep "${1:-delta_lds16}" | cut -c1-400
GPU_MAX_HW_QUEUES=20 timeout 200 python tools/fromhost_ab.py 3000 5 4 resident,pinned,resident,pinned 2>&1 | grep -v "amdgpu\|corto_hip pool" | cut -c1-150
