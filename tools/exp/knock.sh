#!/bin/bash
# GPU box: tools/exp/knock.py for a list of knock-outs -> gpurun_out/knock.txt
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/knock.txt
for k in "$@"; do
  echo "== $k"
  timeout -k 10 150 python3 tools/exp/knock.py "$k" --steps 200 --warmup 20 > gpurun_out/knock_one.json 2> gpurun_out/knock_one.err || { tail -5 gpurun_out/knock_one.err; exit 1; }
  python3 -c "
import json
for l in open('gpurun_out/knock_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('%-40s %.1f frames/s' % ('$k', d['value']))
" | tee -a gpurun_out/knock.txt
done
