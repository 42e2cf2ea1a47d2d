#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace of the default bench run + PMC passes of the headline step at 10 / 20 / 50 %.
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
: > $OUT/status.txt
# 1. kernel trace of a bench run with the extras (shorter than the default 200 steps; the averages do not depend on it)
DB=$(run_prof trace --kernel-trace --stats -- python $REPO/bench.py --steps 60 --warmup 10 --no-cpu-baseline)
[ -n "$DB" ] && python $REPO/scripts/rocpd_stats.py $DB $OUT/kernel_trace.md > /dev/null
tail -1 /tmp/rp_trace/log.txt > $OUT/bench_line_profiled.json
# 2. PMC passes per density (separate passes: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2)
for D in 0.1 0.2 0.5; do
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    tag="pmc_d${D}_$(echo $P | cut -d' ' -f1)"
    DB=$(run_prof $tag --kernel-trace --pmc $P -- python $REPO/bench.py --density $D --steps 8 --warmup 2 --prewarm-seconds 0 --no-extras --no-cpu-baseline)
    echo "## density $D pass: $P" >> $OUT/pmc.txt
    [ -n "$DB" ] && python $REPO/scripts/rocpd_pmc.py $DB bsmm >> $OUT/pmc.txt 2>&1
  done
done
cat $OUT/status.txt
