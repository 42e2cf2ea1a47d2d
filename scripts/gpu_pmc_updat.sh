#!/bin/bash
# PMC passes over scripts/gpu_updat_one.py (AXIS from the environment, one case) -> gpurun_out/pmc_updat_<TAG>.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
CASE=${1:-d20:auto}
OUT=$REPO/gpurun_out/pmc_updat_${TAG:-default}.txt
mkdir -p $REPO/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
         "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/rp_u$i; mkdir -p /tmp/rp_u$i; cd /tmp/rp_u$i
  timeout 170 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_updat_one.py $CASE > log.txt 2>&1
  echo "## pass $i rc=$?: $P" >> $OUT
  DB=$(find /tmp/rp_u$i -name "*results.db" | head -1)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB updat32_a1_v2 >> $OUT 2>&1
done
tail -5 $OUT
