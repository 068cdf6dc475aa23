#!/bin/bash
# GPU box: throughput against lanes / depth / workers
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "6 18 12" "8 24 12" "10 32 12" "6 18 10" "6 24 12"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu --steps 240 --warmup 12 --lanes $1 --depth $2 --workers $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $1 depth $2 workers $3', round(d['value'],1), d['pipeline_wall_ms_per_frame'])" || exit 1
done
