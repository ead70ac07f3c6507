# A/B of the spatially partitioned forward pass (OAK_PARTITION=0 disables it; OAK_PART_CUS = the side stream's compute units):
# row shards of the headline problem and C2
set -u
ROOT=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$ROOT"
O=gpurun_out/r05; mkdir -p $O
TAG=${1:-ab}
run() {
  python3 tools/dev_shard.py 4 8 16 2>&1 | tail -3
  python3 bench.py --config c2 --steps 200 --warmup 20 --no-cpu-baseline --no-fit 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c2 ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phase_ms_per_step'].items()}, 'fwd+grad %.3f' % d.get('forward_plus_gradient',{}).get('ms_per_step'), 'whitened %.3f' % d.get('whitened',{}).get('ms_per_step'))"
}
{
echo "== OAK_PARTITION=0"; OAK_PARTITION=0 run
for cus in ${CUS:-16 32}; do echo "== partition, OAK_PART_CUS=$cus"; OAK_PART_CUS=$cus run; done
} 2>&1 | tee $O/${TAG}_partition.txt
