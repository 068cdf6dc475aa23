#!/bin/bash
# GPU box: headline rate beside N spinning processes, for environment settings of the bench process (e.g. OMP_NUM_THREADS=1)
# usage: hog_env.sh NHOGS "VAR=val ..." ...    ("-" = no setting)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/hog_env.txt
nh=$1; shift
pids=""
for i in $(seq $nh); do python3 -c "
import time
t=time.time()
while time.time()-t < 200: pass
" & pids="$pids $!"; done
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e timeout -k 10 120 python3 bench.py --no-cpu --no-extras --steps 300 --warmup 20 > gpurun_out/hog_one.json 2> gpurun_out/hog_one.err || { tail -5 gpurun_out/hog_one.err; kill $pids; exit 1; }
  python3 -c "
import json
for l in open('gpurun_out/hog_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('hogs %s env [%s] %.1f frames/s  (zogy group %.2f ms, latency %.1f ms, lane cpu %.1f ms)' % ('$nh', '$cfg', d['value'], d['roofline']['avg_launch_ms'], d['single_frame_latency_ms'], d['host_ms_per_frame']['lane_threads_cpu']))
" | tee -a gpurun_out/hog_env.txt
done
kill $pids 2>/dev/null
