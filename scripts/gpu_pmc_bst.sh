#!/bin/bash
# counters of one attention operator at BASELINE configs[4]: OP=nt|fused|sm|nn PAT=<kernel name substring> scripts/gpu_pmc_bst.sh -> gpurun_out/pmc_bst_$OP.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/pmc_bst_${OP:-nt}.txt
mkdir -p $REPO/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_CYCLES" \
         "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/rp_b$i; mkdir -p /tmp/rp_b$i; cd /tmp/rp_b$i
  timeout 170 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_bst_one.py > log.txt 2>&1
  echo "## pass $i rc=$?: $P" >> $OUT
  DB=$(find /tmp/rp_b$i -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $REPO/scripts/rocpd_pmc.py $DB ${PAT:-bst_nt} >> $OUT 2>&1; else tail -5 log.txt >> $OUT; fi
done
tail -3 $OUT
