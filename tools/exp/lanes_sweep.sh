#!/bin/bash
# GPU box: throughput against the number of stage-C lanes / frames in flight
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "2 12" "3 12" "4 16" "6 18"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu --steps 160 --warmup 8 --lanes $1 --depth $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $1 depth $2', round(d['value'],1), d['pipeline_wall_ms_per_frame'])" || exit 1
done
