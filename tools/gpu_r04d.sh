#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r04d
timeout 900 python -m pytest tests -m gpu -q > ${O}_suite.txt 2>&1; echo "suite rc $?"; tail -8 ${O}_suite.txt
( timeout 200 python tools/fromhost_ab.py 3000 4 4 pinned,pinned-next-kernel,pinned-head-kernel,pinned-next-dma,resident
  timeout 200 python tools/fromhost_ab.py 3000 5 4 pinned,pinned-next-kernel,pinned-head-kernel
  timeout 200 python tools/fromhost_ab.py 3000 4 4 scattered,scattered-next-kernel ) > ${O}_fromhost_ab.txt 2>&1
grep -v amdgpu.ids ${O}_fromhost_ab.txt
