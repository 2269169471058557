#!/bin/bash
# what bounds the pipelined rate: the same pool with kernels asking for MORE LDS than they use (nothing else changes); KIND=irregular: flipped diagonals
cd $GRAFT_REPO_ROOT
for v in "" "CORTO_EXP_LDS_PAD_DELTA=20" "CORTO_EXP_LDS_PAD_TOPO=12" "CORTO_EXP_LDS_PAD_TOPO=24" "CORTO_EXP_LDS_PAD_NORMAL=20"; do
  echo -n "$v : "; env $v python tools/shape_probe.py "4 4" 2>&1 | tail -1
done
