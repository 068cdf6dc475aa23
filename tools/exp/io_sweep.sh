#!/bin/bash
# GPU box: files-to-files rate against writer / reader threads
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4e/io_sweep.log; mkdir -p gpurun_out/r4e; : > $out
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --io-only --steps 40 --writers $1 --readers $2 --lanes $3 --depth $4 > /tmp/io.json 2>/tmp/io.err
  python3 - "$cfg" >> $out <<PY
import json,sys
try:
    d=json.loads(open("/tmp/io.json").read())
    for k in ("ramdisk","scratch"):
        r=d[k]
        print(sys.argv[1], k, "fps %.1f" % r["frames_per_s"], "out-only %.1f" % r["output_side_only"]["frames_per_s"], r["host_ms_per_frame"])
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/io.err").read()[-500:])
PY
done
cat $out
