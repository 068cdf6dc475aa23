#!/bin/bash
# GPU box: headline rate against lanes / frames in flight (A / B / A / B)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4j/lanes.log; mkdir -p gpurun_out/r4j; : > $out
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps ${STEPS:-150} --warmup 20 --no-extras --no-cpu --lanes $1 --depth $2 > /tmp/b.json 2>/tmp/b.err
  python3 - "$cfg" >> $out <<PY
import json,sys
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps %.1f" % d["value"], "idle_to_idle %.1f" % d["idle_to_idle"]["frames_per_s"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat $out
