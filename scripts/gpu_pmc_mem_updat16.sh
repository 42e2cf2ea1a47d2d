#!/bin/bash
# memory-system counters of the bsize-16 / feature axis 0 weight gradient at BASELINE configs[2]: the row-owner kernel, then the windowed one
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for MODE in rows win; do
  OUT=$REPO/gpurun_out/pmc_mem_updat16_$MODE.txt; : > $OUT
  [ $MODE = win ] && export WINDOWED=1 || unset WINDOWED
  i=0
  while read -r P; do
    [ -z "$P" ] && continue
    i=$((i+1)); rm -rf /tmp/rp_u$i; mkdir -p /tmp/rp_u$i; cd /tmp/rp_u$i
    XP_REPS=6 timeout 120 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_updat16_one.py > log.txt 2>&1
    echo "## pass: $P (rc=$?)" >> $OUT
    DB=$(find /tmp/rp_u$i -name "*results.db" | head -1)
    [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB "updat16" >> $OUT 2>&1
  done <<'LIST'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
LIST
done
cat $REPO/gpurun_out/pmc_mem_updat16_rows.txt $REPO/gpurun_out/pmc_mem_updat16_win.txt
