#!/bin/bash
# GPU box: headline rate with busy host cores (N spinning processes beside the bench), against lanes / frames in flight:
# how much host slack the pipeline has.   usage: hog_lanes.sh NHOGS "lanes depth" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/hog.txt
nh=$1; shift
pids=""
for i in $(seq $nh); do python3 -c "
import time
t=time.time()
while time.time()-t < 170: pass
" & pids="$pids $!"; done
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 120 python3 bench.py --no-cpu --no-extras --steps 300 --warmup 20 --lanes $1 --depth $2 > gpurun_out/hog_one.json 2> gpurun_out/hog_one.err || { tail -5 gpurun_out/hog_one.err; kill $pids; exit 1; }
  python3 -c "
import json
for l in open('gpurun_out/hog_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('hogs %s lanes/depth %-6s %.1f frames/s  (zogy group %.2f ms, latency %.1f ms)' % ('$nh', '$cfg', d['value'], d['roofline']['avg_launch_ms'], d['single_frame_latency_ms']))
" | tee -a gpurun_out/hog.txt
done
kill $pids 2>/dev/null
