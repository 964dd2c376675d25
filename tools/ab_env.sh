#!/bin/bash
# Same-session A/B of one environment switch of the library / engine on one GPU box: alternating bench runs, it/s per run.
#   bash tools/ab_env.sh "<ENV=a ...>" "<ENV=b ...>" [rounds] [extra bench args...]
A=$1; B=$2; N=${3:-3}; shift 3 || true
R=$GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for v in A B; do
    e=$A; [ $v = B ] && e=$B
    env $e timeout -k 10 200 python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-modes --no-sampling --no-extras "$@" 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$e', round(d['value'],1), round(d['roofline']['avg_us_per_launch'],3), d['roofline']['launches_timed'])" || exit 1
  done
done
