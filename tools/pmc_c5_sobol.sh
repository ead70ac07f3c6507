#!/bin/bash
# PMC passes (separate from any kernel trace) for config 5's Sobol pass: fabric traffic and matrix-pipe busy cycles of the
# index-pair SYRK (syrk_kernel<32, true>) and the panel builder.  usage: tools/pmc_c5_sobol.sh <tag>
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_c5_sobol_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  NAME=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline --no-fit > /dev/null 2> $OUT/pmc_$NAME.err
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - $OUT/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["counters_per_launch"]
for k, v in d.items():
    if "sobol" in k or "syrk_kernel<32, true>" in k:
        print(k[:60], {c: f"{x:.4g}" for c, x in v.items()})
PY
rm -rf $OUT/pmc_*/*/*.db
