#!/bin/bash
# GPU box: throughput against host-pool message size and worker count
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "1 12" "2 12" "4 12" "4 13" "4 14" "8 14" "2 14"; do
  set -- $cfg
  BBX_HOST_CHUNK=$1 timeout -k 10 200 python bench.py --no-cpu --steps 160 --warmup 8 --workers $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $1 workers $2', round(d['value'],1), d['pipeline_wall_ms_per_frame'])" || exit 1
done
