#!/bin/bash
# What co-resident automata cost each other: PMC of k_topology_lds on ONE batch of $NB blobs (256: a wave a CU; 2048: eight), $MESH as tools/kt_probe_irregular.py.
# usage: MESH=delaunay NB=2048 bash tools/prof_topo_scale.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_topo_scale
rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
i=0
SETS=("SPI_RA_LDS_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_RES_STALL_CSN SPI_RA_REQ_NO_ALLOC_CSN"
      "SPI_RA_WVLIM_STALL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_BULKY_CU_FULL_CSN SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES")
if [ -z "$SPI_ONLY" ]; then SETS+=("SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
      "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_VMEM_WR"
      "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_SALU"); fi
for SET in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --pmc $SET -d $OUT/p$i -o p -- python tools/kt_probe_irregular.py > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_topology_lds" in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("NB", "$NB", "MESH", "$MESH", {k: round(v) for k, v in sorted(tot.items())})
PY
grep -h "topology_lds" $OUT/log1.txt | cut -c1-200
rm -rf $OUT/p*
