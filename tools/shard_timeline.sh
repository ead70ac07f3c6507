# kernel timeline of one forward step on an N/8 row shard of the headline problem (both streams)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out/shardtl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shardtl -o s8 -- python3 tools/dev_shard.py 8 > gpurun_out/shardtl/run.log 2>&1
tail -1 gpurun_out/shardtl/run.log
