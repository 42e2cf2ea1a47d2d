#!/bin/bash
# PMC passes over scripts/gpu_xrows_one.py (one density) -> gpurun_out/pmc_xrows_$TAG.txt.  $1 = density, BSMM_LIB / FLOW honoured.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
D=${1:-20}
PAT=${PAT:-xrows32}
OUT=$REPO/gpurun_out/pmc_xrows_${TAG:-default}.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" \
         "SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/rp_x$i; mkdir -p /tmp/rp_x$i; cd /tmp/rp_x$i
  XP_REPS=6 timeout 170 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_xrows_one.py $D > log.txt 2>&1
  echo "## pass $i rc=$?: $P" >> $OUT
  DB=$(find /tmp/rp_x$i -name "*results.db" | head -1)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB $PAT >> $OUT 2>&1
done
tail -5 $OUT
