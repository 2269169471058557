#!/bin/bash
# build a VARIANT of the library beside the product for an A/B measurement: tools/ab_build.sh <name> <DEFINE[,DEFINE...]>
# -> corto_amd/lib_<name>/libcorto_hip.so; run a probe against it with CORTO_HIP_LIB_PATH=corto_amd/lib_<name>/libcorto_hip.so
cd /root/repo || exit 1
CORTO_BUILD_LIBDIR=/root/repo/corto_amd/lib_$1 CORTO_BUILD_DEFINES=$2 python -m corto_amd.build > /tmp/ab_build_$1.log 2>&1 || { tail /tmp/ab_build_$1.log; exit 1; }
ls -la corto_amd/lib_$1/libcorto_hip.so
