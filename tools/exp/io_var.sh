#!/bin/bash
# GPU box: files-to-files rate with knock-out builds of the library (timing only: their products are wrong)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4j; out=gpurun_out/r4j/io_var.log; : > $out
for v in product "$@" product; do
  if [ "$v" = product ]; then e=(A=1); else e=(BBX_LIB_PATH=tools/exp/_var/$v/libbbx_hip.so); fi
  env "${e[@]}" timeout -k 10 300 python bench.py --io-only --io-simple --steps 60 > /tmp/io.json 2>/tmp/io.err
  python3 - "$v" >> $out <<PY
import json,sys
try:
    r=json.loads(open("/tmp/io.json").read())["ramdisk"]
    print(sys.argv[1], "fps %.1f" % r["frames_per_s"], r.get("error",""))
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/io.err").read()[-300:])
PY
done
cat $out
