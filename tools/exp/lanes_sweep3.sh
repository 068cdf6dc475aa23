#!/bin/bash
# GPU box: headline rate and the live duration of the ZOGY launch group against the number of lanes
cd "$GRAFT_REPO_ROOT" || exit 1
for l in 2 3 4 5 6 8; do
  timeout -k 10 200 python3 bench.py --no-cpu --no-extras --lanes $l --depth $((l * 3 > 8 ? l * 3 : 8)) > gpurun_out/ls_$l.json 2> gpurun_out/ls_$l.err || exit 1
  echo "lanes $l: $(python3 tools/dbg/bench_sum.py gpurun_out/ls_$l.json | head -2 | tr '\n' ' ')"
done
