#!/bin/bash
# GPU box: throughput with LA-Cosmic's background level fed in advance (all frames) vs on demand
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2 3; do for v in 0 100000; do
  BBX_LAC_FEED_FRAMES=$v python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feed_frames=$v', round(d['value'],1))"
done; done
