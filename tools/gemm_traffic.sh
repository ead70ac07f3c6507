#!/bin/bash
# Fabric traffic (FETCH_SIZE, doubled per the guide) and time of the backward adjoint GEMM under the two workgroup-to-XCD mappings
# (OAK_GEMM_XCD_COLS=0: all eight column tiles of a row block on one XCD; 2: two column tiles per XCD).  usage (GPU box): tools/gemm_traffic.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gemm_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in ${VS:-0 2}; do
  export OAK_GEMM_XCD_COLS=$V
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$V -- python3 $ROOT/tools/dev_bwd_time.py > /dev/null 2> $OUT/pmc_$V.err
  for r in 1 2; do timeout 200 python3 $ROOT/tools/dev_bwd_time.py 2>&1 | head -1 >> $OUT/time_$V.txt; done
  python3 - $OUT $V <<'PY'
import sys, glob, csv, collections
out, v = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/pmc_{v}/*/*counter_collection.csv")[0]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and int(r["Grid_Size"]) > 1_000_000:
        k = r["Kernel_Name"].split("(")[0][:48]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in tot:
    if "gemm128" in k: print(f"XCD_COLS={v}: {k}: FETCH_SIZE x2 = {2 * 1024 * tot[k] / n[k] / 1e9:.2f} GB per launch ({n[k]} launches)")
for l in open(f"{out}/time_{v}.txt"): print(f"XCD_COLS={v}:", l.strip()[:250])
PY
done
rm -rf $OUT/pmc_*/*/*.db
