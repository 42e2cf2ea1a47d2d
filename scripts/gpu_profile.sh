#!/bin/bash
# rocprofv3 evidence for profiles/: per WORKLOAD (headline at 10 / 20 / 50 %, BASELINE configs[2], configs[3]) one kernel trace
# and the PMC passes (separate runs: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2), plus the unprofiled default bench line.
# Runs on the GPU box (gpurun); every profiler invocation sits under its own `timeout`.  Output: gpurun_out/prof/*.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_prof() {   # $1 = tag, $2.. = rocprofv3 options ; then -- command
  local tag=$1; shift
  rm -rf /tmp/rp_$tag; mkdir -p /tmp/rp_$tag; cd /tmp/rp_$tag
  timeout 170 rocprofv3 "$@" > /tmp/rp_$tag/log.txt 2>&1
  echo "[$tag] rc=$?" >> $OUT/status.txt
  DB=$(find /tmp/rp_$tag -name "*results.db" | head -1)
  echo "$DB"
}
: > $OUT/status.txt; : > $OUT/pmc.txt
python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
WORKLOADS=("d10|--density 0.1" "d20|--density 0.2" "d50|--density 0.5" "cfg2|--config cfg2" "cfg3|--config cfg3")
for W in "${WORKLOADS[@]}"; do
  KEY=${W%%|*}; ARGS=${W#*|}
  # 1. kernel trace of this workload alone (so that "avg us" is one density's, not a mix)
  DB=$(run_prof trace_$KEY --kernel-trace --stats -- python $REPO/bench.py $ARGS --no-extras --steps 100 --warmup 20)
  [ -n "$DB" ] && python $REPO/scripts/rocpd_stats.py $DB $OUT/kernel_trace_$KEY.md > /dev/null
  tail -1 /tmp/rp_trace_$KEY/log.txt > $OUT/bench_line_profiled_$KEY.json
  # 2. PMC passes
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    tag="pmc_${KEY}_$(echo $P | cut -d' ' -f1)"
    DB=$(run_prof $tag --kernel-trace --pmc $P -- python $REPO/bench.py $ARGS --steps 8 --warmup 2 --prewarm-seconds 0 --no-extras)
    echo "## workload $KEY pass: $P" >> $OUT/pmc.txt
    [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB bsmm >> $OUT/pmc.txt 2>&1
  done
done
cat $OUT/status.txt
