#!/bin/bash
# GPU box: headline rate against the runtime's number of hardware queues (GPU_MAX_HW_QUEUES), interleaved repeats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/hwq.txt
rounds=$1; shift
for r in $(seq $rounds); do
  for q in "$@"; do
    if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout -k 10 150 python3 bench.py --no-cpu --no-extras --steps 300 --warmup 20 ${BENCH_ARGS} > gpurun_out/hwq_one.json 2> gpurun_out/hwq_one.err || { tail -5 gpurun_out/hwq_one.err; exit 1; }
    python3 -c "
import json
for l in open('gpurun_out/hwq_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('hwq %-8s %.1f frames/s' % ('$q', d['value']))
" | tee -a gpurun_out/hwq.txt
  done
done
