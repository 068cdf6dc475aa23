#!/bin/bash
# GPU box: headline rate, product library against scratch libraries (tools/exp/_var/<name>/libbbx_hip.so), interleaved
# usage: tools/exp/ab.sh ROUNDS name [name ...]      ("product" = the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/ab.txt
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = product ]; then unset BBX_LIB_PATH; else export BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so; fi
    timeout -k 10 150 python3 bench.py --no-cpu --no-extras --steps 300 --warmup 20 ${BENCH_ARGS} > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { tail -5 gpurun_out/ab_one.err; exit 1; }
    python3 -c "
import json
for l in open('gpurun_out/ab_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('%-12s %.1f frames/s' % ('$v', d['value']))
" | tee -a gpurun_out/ab.txt
  done
done
