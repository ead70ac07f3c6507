#!/bin/bash
# SYRK fabric traffic and time against the number of row splits (OAK_SYRK_NSPLIT): one FETCH_SIZE pass + one timing run each.
# usage (on the GPU box): tools/syrk_traffic.sh "128 192 256"
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
OUT=$ROOT/gpurun_out/syrk_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NS in ${1:-128 256}; do
  export OAK_SYRK_NSPLIT=$NS
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$NS -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit --route phi > /dev/null 2> $OUT/pmc_$NS.err
  python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fit > $OUT/bench_$NS.json 2> /dev/null
  python3 - $OUT $NS <<'PY'
import sys, glob, csv, json, collections
out, ns = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/pmc_{ns}/*/*counter_collection.csv")[0]
tot = collections.defaultdict(float); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].split("(")[0][:40]
        tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
d = json.loads(open(f"{out}/bench_{ns}.json").read().strip().splitlines()[-1])
for k, v in tot.items():
    if "syrk_kernel" in k or "syrk_reduce" in k:
        print(f"nsplit {ns}: {k}: FETCH_SIZE x2 = {2 * 1024 * v / len(disp[k]) / 1e9:.2f} GB per launch ({len(disp[k])} launches)")
p = d["phase_ms_per_step"]
print(f"nsplit {ns}: step {d['ms_per_step']:.3f} ms  syrk {p['syrk']:.3f}  reduce {p['reduce']:.3f}  gram {p['gram']:.3f}")
PY
done
rm -rf $OUT/pmc_*/*/*.db
