set -eu
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out/c2tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c2tl -o c2 -- python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c2tl/bench.log 2>&1
tail -1 gpurun_out/c2tl/bench.log | cut -c1-200
f=$(find gpurun_out/c2tl -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py $f > gpurun_out/c2tl/timeline.txt
wc -l gpurun_out/c2tl/timeline.txt
