#!/bin/bash
# Round profiles (run on the GPU box from the repo root): steady-state kernel tables and the two PMC passes per variant.
#   bash scripts/collect_profiles.sh r03                      -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
#   VARIANTS="B" MODES="fwd" PMC="" bash scripts/collect_profiles.sh quick   (a subset: variants / train|fwd / PMC variants)
# rocprofv3 gets the interpreter itself after `--` (no env / bash -c hop), counters in their own passes.
set -o pipefail
TAG=${1:-r03}
VARIANTS=${VARIANTS-"B A H L M"}
MODES=${MODES-"train fwd"}
PMC=${PMC-"B A"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in $VARIANTS; do
  for MODE in $MODES; do
    [ $MODE = fwd ] && [ $V != B ] && [ $V != A ] && continue
    D=$O/${TAG}_${MODE}_${V}
    EXTRA=""; [ $MODE = fwd ] && EXTRA="--fwd-only --dump-levels $O/${TAG}_${V}_levels.json"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 $R/scripts/prof_step.py --variant $V --steps 5 $EXTRA > $D.log 2>&1 || exit 1
    python3 $R/scripts/prof_summary.py steady $D $O/${TAG}_${MODE}_${V}_bs64_summary.md "$MODE step, variant $V, batch 64, 256x256 (rocprofv3 --kernel-trace --stats, steady state)" 5 > /dev/null || exit 1
    cp $D/t_kernel_stats.csv $O/${TAG}_${MODE}_${V}_bs64_kernel_stats.csv
    [ $MODE = fwd ] && { python3 $R/scripts/prof_summary.py levels $D $O/${TAG}_${V}_levels.json $O/${TAG}_fwd_${V}_bs64_levels.md "forward of variant $V per resolution level, batch 64, 256x256" > /dev/null || echo "levels table failed for $V"; }
    echo "$TAG $MODE $V done"
  done
done
for V in $PMC; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$O/${TAG}_pmc_${C}_${V}
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $D -o t --output-format csv -- python3 $R/scripts/prof_step.py --variant $V --steps 4 --fwd-only > $D.log 2>&1 || exit 1
    echo "$TAG pmc $C $V done"
  done
done
find $O -name "*.db" -delete
find $O -name "*_kernel_trace.csv" -size +20M -delete
