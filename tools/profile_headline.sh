#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + separate PMC passes of the headline bench, summaries into gpurun_out/.
# usage: tools/profile_headline.sh <tag>   (e.g. r01_e)
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-fit"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $SET | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit > /dev/null 2> $OUT/pmc_$NAME.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 tools/make_profile_json.py $OUT/summary.json $TAG $OUT      # -> $OUT/traffic.json, $OUT/pmc_counts.json (copy into profiles/)
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/*/*.db
tail -1 $OUT/bench.json
