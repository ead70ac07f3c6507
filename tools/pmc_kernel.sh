#!/bin/bash
# Per-launch PMC counters of the kernels whose name contains PATTERN, one rocprofv3 --pmc pass per counter group (separate passes:
# TCC and SQ groups do not fit one), largest-grid launches only.  FETCH_SIZE is reported in KB and doubled (gfx950 counts 128-byte
# requests as 64 B, MI355X_MICROARCH.md); WRITE_SIZE in KB.
# usage (on the GPU box):  tools/pmc_kernel.sh OUTNAME PATTERN "GROUP1;GROUP2;..." -- python3 script.py args...
#   e.g. tools/pmc_kernel.sh crt_syrk crt_syrk "FETCH_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" -- python3 tools/dev_crt.py --configs headline
set -eu
NAME=$1; PATTERN=$2; GROUPS_=$3; shift 3
[ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
OUT=$ROOT/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GR <<< "$GROUPS_"
i=0
for g in "${GR[@]}"; do
  ( cd $ROOT && rocprofv3 --pmc $g --output-format csv -d $OUT/pass_$i -- "$@" > $OUT/pass_$i.out 2> $OUT/pass_$i.err ) || { tail -5 $OUT/pass_$i.err; }
  i=$((i + 1))
done
python3 - $OUT "$PATTERN" <<'PY' | tee $OUT/summary.json
import sys, glob, csv, json, collections
out, pat = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for f in glob.glob(f"{out}/pass_*/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    big = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]; big[k] = max(big[k], int(r["Grid_Size"]))
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if int(r["Grid_Size"]) != big[k]: continue
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for (k, c), v in acc.items():
        v /= max(1, len(disp[(k, c)]))
        if c == "FETCH_SIZE": res[k]["FETCH_bytes_x2"] = v * 2048.0
        elif c == "WRITE_SIZE": res[k]["WRITE_bytes"] = v * 1024.0
        else: res[k][c] = v
        res[k]["launches"] = len(disp[(k, c)])
json.dump(res, sys.stdout, indent=1, sort_keys=True)
PY
rm -rf $OUT/pass_*/*/*.db
