#!/bin/bash
# GPU box: headline rate against lanes / frames in flight (two runs each)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "6 16" "6 24" "7 21" "5 15" "6 12"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu --no-extras --lanes $1 --depth $2 > gpurun_out/ls.json 2> gpurun_out/ls.err || exit 1
    echo "lanes $1 depth $2: $(python3 tools/dbg/bench_sum.py gpurun_out/ls.json | head -1)"
  done
done
