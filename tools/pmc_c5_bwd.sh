#!/bin/bash
# PMC passes for the backward pair kernel of config C5 (D = 32 mixed): instruction mix and wait cycles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  NAME=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$NAME -- python3 $ROOT/bench.py --config c5 --grad --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/$NAME.err
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_c5/*/*/*counter_collection.csv') + glob.glob('gpurun_out/pmc_c5/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'gram_bwd_fast_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items(): out[k] = (v / max(n, 1), n)
for k, (v, n) in out.items(): print(f"{k:28s} {v:.4e} per launch ({n} launches)")
PY
