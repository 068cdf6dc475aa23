#!/bin/bash
# GPU box: files-to-files rate with the runtime's copy paths switched (SDMA engines on / off)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4i/sdma.log; mkdir -p gpurun_out/r4i; : > $out
for cfg in "default" "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0" "default"; do
  if [ "$cfg" = "default" ]; then e=(); else e=("$cfg"); fi
  env "${e[@]}" timeout -k 10 300 python bench.py --io-only --io-simple --steps 60 > /tmp/io.json 2>/tmp/io.err
  python3 - "$cfg" >> $out <<PY
import json,sys
try:
    r=json.loads(open("/tmp/io.json").read())["ramdisk"]
    print(sys.argv[1], "fps %.1f" % r["frames_per_s"], r["host_ms_per_frame"], {k: round(v["wall"],1) for k,v in r["writer_ms_per_image"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/io.err").read()[-300:])
PY
done
cat $out
