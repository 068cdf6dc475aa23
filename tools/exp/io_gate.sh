#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4k; out=gpurun_out/r4k/io_gate.log; : > $out
for cfg in "A=1" "BBX_ZOGY_GATE=0" "A=1" "BBX_ZOGY_GATE=0"; do
  env $cfg timeout -k 10 300 python bench.py --io-only --io-simple --steps 80 > /tmp/io.json 2>/tmp/io.err
  python3 - "$cfg" >> $out <<PY
import json,sys
try:
    r=json.loads(open("/tmp/io.json").read())["ramdisk"]
    print(sys.argv[1], "io fps %.1f %s" % (r["frames_per_s"], r.get("error","")))
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/io.err").read()[-300:])
PY
done
cat $out
