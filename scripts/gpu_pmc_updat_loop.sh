#!/bin/bash
# SQ counters of the streaming weight-gradient kernel for the library in BSMM_LIB (default build when unset): where the wave cycles of an
# interval go.  TAG=name scripts/gpu_pmc_updat_loop.sh [case] -> gpurun_out/pmc_loop_<TAG>.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
CASE=${1:-d20:auto}
OUT=$REPO/gpurun_out/pmc_loop_${TAG:-default}.txt
mkdir -p $REPO/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU_MFMA_BF16 SQ_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/rp_l$i; mkdir -p /tmp/rp_l$i; cd /tmp/rp_l$i
  timeout 170 rocprofv3 --kernel-trace --pmc $P -- python $REPO/scripts/gpu_updat_one.py $CASE > log.txt 2>&1
  echo "## pass $i rc=$?: $P" >> $OUT
  DB=$(find /tmp/rp_l$i -name "*results.db" | head -1)
  if [ -n "$DB" ]; then python $REPO/scripts/rocpd_pmc.py $DB updat32_a1_v2 >> $OUT 2>&1; else tail -5 log.txt >> $OUT; fi
done
tail -3 $OUT
