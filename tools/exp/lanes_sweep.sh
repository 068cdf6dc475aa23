#!/bin/bash
# GPU box: headline rate against lanes / frames in flight / wait sleep
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4e/sweep.log; mkdir -p gpurun_out/r4e; : > $out
for cfg in "50 6 16" "20 6 16" "50 8 20" "20 8 20" "50 10 24" "0 6 16"; do
  set -- $cfg
  BBX_WAIT_SLEEP_US=$1 timeout -k 10 200 python bench.py --steps 60 --warmup 6 --no-extras --no-cpu --lanes $2 --depth $3 > /tmp/b.json 2>/tmp/b.err
  python3 - "$cfg" >> $out <<PY
import json,sys
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps %.1f" % d["value"], d["host_ms_per_frame"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat $out
