#!/bin/bash
# build + CPU suite here, then the GPU call: a stale .so or a red CPU test must not cost GPU minutes
# usage: tools/gpurun_checked.sh <timeout_s> '<command>'
cd /root/repo || exit 1
python -m corto_amd.build > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; echo "BUILD FAILED"; exit 1; }
make -s -C oracle all || exit 1
timeout 900 python -m pytest tests -x -q -m "not gpu" > /tmp/cpu_suite.log 2>&1 || { tail -20 /tmp/cpu_suite.log; echo "CPU SUITE FAILED"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
