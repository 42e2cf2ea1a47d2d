#!/bin/bash
# memory-system counters of one pass of the bench shape: SIDE=bprop DENS=20 XP_OPT=0 scripts/gpu_pmc_mem.sh  ->  gpurun_out/r3/pmc_mem_<tag>.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
SIDE=${SIDE:-bprop}; DENS=${DENS:-20}; TAG=${TAG:-$SIDE$DENS}
OUT=$REPO/gpurun_out/r3/pmc_mem_$TAG.txt
mkdir -p $REPO/gpurun_out/r3; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1)); rm -rf /tmp/rp_m$i; mkdir -p /tmp/rp_m$i; cd /tmp/rp_m$i
  timeout 120 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_bprop_loop.py $DENS $SIDE 12 > log.txt 2>&1
  echo "## pass: $P (rc=$?)" >> $OUT
  DB=$(find /tmp/rp_m$i -name "*results.db" | head -1)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB "${KERNEL:-xcol32}" >> $OUT 2>&1
done <<'LIST'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
LIST
cat $OUT
