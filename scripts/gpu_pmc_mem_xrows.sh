#!/bin/bash
# memory-system counters of the row-split kernel (or FLOW=1: the flow kernel) at the bench shape: TAG=.. BSMM_LIB=.. scripts/gpu_pmc_mem_xrows.sh 20
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
D=${1:-20}
PAT=${PAT:-xrows32}
OUT=$REPO/gpurun_out/pmc_mem_xrows_${TAG:-default}.txt
mkdir -p $REPO/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r P; do
  [ -z "$P" ] && continue
  i=$((i+1)); rm -rf /tmp/rp_m$i; mkdir -p /tmp/rp_m$i; cd /tmp/rp_m$i
  XP_REPS=6 timeout 120 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_xrows_one.py $D > log.txt 2>&1
  echo "## pass: $P (rc=$?)" >> $OUT
  DB=$(find /tmp/rp_m$i -name "*results.db" | head -1)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB "$PAT" >> $OUT 2>&1
done <<'LIST'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
LIST
cat $OUT
