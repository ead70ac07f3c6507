#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the stand-alone triangular solve.  usage: tools/profile_trsm.sh <tag> [M] [rows]
set -eu
TAG=${1:-rXX}; M=${2:-1024}; ROWS=${3:-524288}
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
OUT=$ROOT/gpurun_out/trsm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/dev_trsm_time.py $M $ROWS 3 > $OUT/time.txt 2> $OUT/time.err || true
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $SET | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$NAME -- python3 $ROOT/tools/dev_trsm_time.py $M $ROWS 1 > /dev/null 2> $OUT/pmc_$NAME.err || true
done
cd $ROOT
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/*/*.db
cat $OUT/time.txt
python3 - <<PY
import json
s=json.load(open("$OUT/summary.json"))
for k,v in s["kernels"].items():
    if "trsm" in k or "gemm128" in k: print(k, v)
for k,v in s["counters_per_launch"].items():
    if "trsm_fused" in k or "gemm128" in k: print(k, json.dumps(v))
PY
