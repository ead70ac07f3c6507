#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace of the config-5 bench (N = 262 144, D = 32 mixed, M = 2048, depth 4) including its
# Sobol pass (sobol_panel_kernel + syrk_kernel over the index pairs), kernel-stats CSV + the bench line into gpurun_out/.
# usage: tools/profile_c5.sh <tag>
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_c5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-fit > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace/*/*.db
head -25 $OUT/kernel_stats.csv
