#!/bin/bash
# Repeats the N/8 row-shard evaluation loop (the partitioned forward pass) in fresh processes; a run that exceeds LIMIT seconds gets its
# native stacks dumped through a debugger (started from this shell, by PID) before it is killed (by PID).  usage: tools/soak_shard.sh [runs] [limit]
RUNS=${1:-100}; LIMIT=${2:-60}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/soak_shard; mkdir -p $OUT
ok=0; hung=0
for i in $(seq 1 $RUNS); do
  case $((i % 4)) in 0) export OAK_SYRK_NSPLIT=128;; 1) unset OAK_SYRK_NSPLIT;; 2) export OAK_SYRK_NSPLIT=64;; 3) export OAK_SYRK_NSPLIT=96;; esac
  python3 -u $ROOT/tools/dev_shard.py 8 > $OUT/run.txt 2>&1 &
  pid=$!
  t=0
  while kill -0 $pid 2>/dev/null && [ $t -lt $((LIMIT * 10)) ]; do sleep 0.1; t=$((t + 1)); done
  if kill -0 $pid 2>/dev/null; then
    hung=$((hung + 1))
    echo "run $i (NSPLIT=${OAK_SYRK_NSPLIT:-default}) HUNG; output so far: $(tail -c 200 $OUT/run.txt)"
    /opt/rocm/bin/rocgdb -p $pid -batch -ex "set pagination off" -ex "info sharedlibrary" -ex "thread apply all bt 30" > $OUT/hang_$i.txt 2>&1
    kill -9 $pid; wait $pid 2>/dev/null
  else
    wait $pid; rc=$?
    if [ $rc -ne 0 ]; then echo "run $i rc=$rc: $(tail -c 300 $OUT/run.txt)"; else ok=$((ok + 1)); fi
  fi
done
echo "soak_shard: $ok ok, $hung hung of $RUNS"
