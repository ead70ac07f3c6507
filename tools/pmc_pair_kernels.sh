#!/bin/bash
# Where do the pair kernels' empty issue slots go?  Instruction-mix, busy and WAIT counters (separate rocprofv3 --pmc passes, no
# tracing) of the forward Gram kernel and both backward pair kernels at the headline shape.
# usage (GPU box): tools/pmc_pair_kernels.sh <tag>
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pair_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
SETS=${SETS:-"SQ_INSTS_VALU,SQ_INSTS_SALU SQ_INSTS_LDS,SQ_INSTS_SMEM SQ_WAVE_CYCLES,SQ_BUSY_CYCLES SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_LDS SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU,SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM,SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM,SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT,SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL,SQ_IFETCH_LEVEL"}
for SETC in $SETS; do
  SET=$(echo $SETC | tr ',' ' ')
  NAME=$(echo $SET | tr ' ' '_')
  for ROWS in 0 1; do
    OAK_BWD_ROWS=$ROWS timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/${NAME}_rows$ROWS -- python3 $ROOT/tools/dev_bwd_time.py > /dev/null 2> $OUT/${NAME}_rows$ROWS.err
  done
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv') + glob.glob(out + '/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        key = 'gram_kernel' if 'oak::gram_kernel<' in kn and int(r.get('Grid_Size', '0') or 0) > 4_000_000 else \
              ('bwd_cols' if ('gram_bwd_fast_kernel' in kn and int(r.get('Grid_Size', '0') or 0) > 1_000_000) else ('bwd_rows' if 'gram_bwd_rows_kernel' in kn else None))
        if key is None: continue
        a = acc[(key, r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
    for (k, c), (v, n) in acc.items():
        if (k == 'bwd_cols') == ('_rows0' in f) or k == 'bwd_rows' or (k == 'fwd_gram' and '_rows0' in f): res.setdefault(k, {})[c] = v / max(n, 1)
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v:.4e}")
PY
rm -rf $OUT/*/*/*.db $OUT/*/*.db 2>/dev/null
