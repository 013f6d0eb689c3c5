#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box.  Usage (from the repo root):
#   bash profiles/collect.sh r03
# Writes raw output under gpurun_out/prof_<round>/; profiles/summarize.py turns it into the
# committed summaries.  Counter passes run separately from the kernel trace, one block of
# counters per run.  The traced run leaves out the late-state, N = 50M and per-rank projection legs so
# that the per-kernel averages of the trace are those of the headline state (steps 4-13 after the
# bodies are set; the GPU-tree-mode leg, the direct sum and the 3-D legs are in it).
set -u
R=${1:-r03}
OUT=gpurun_out/prof_$R
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --late-step 0 --no-50m --no-projection > $OUT/bench.json 2> $OUT/bench.err || exit 1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  TAG=$(echo $C | tr ' ' '_' | cut -c1-40)
  # the headline run alone, early (steps 2-3) and late (steps 300-301): summarize.py takes the first
  # three and the last two launches of the traversal kernel apart
  timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$TAG -- python3 bench.py --steps 2 --warmup 1 --headline-only --no-prewarm --late-step 300 --later-step 0 > $OUT/pmc_$TAG.json 2> $OUT/pmc_$TAG.err || exit 1
done
echo collected
