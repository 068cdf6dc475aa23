#!/bin/bash
# GPU box: files-to-files rate with the fpack launches as gated sections (BBX_FPACK_GATE=1) against free-running ones
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; out=gpurun_out/io_fgate.txt; : > $out
for cfg in "$@"; do
  env BBX_FPACK_GATE=$cfg timeout -k 10 300 python bench.py --io-only --io-simple --steps ${IO_STEPS:-100} > gpurun_out/io_one.json 2> gpurun_out/io_one.err || { tail -5 gpurun_out/io_one.err; exit 1; }
  python3 - "$cfg" <<PY | tee -a $out
import json,sys
r=json.loads(open("gpurun_out/io_one.json").read())["ramdisk"]
print("BBX_FPACK_GATE=%s  files-to-files %.1f frames/s %s" % (sys.argv[1], r["frames_per_s"], r.get("error","")))
PY
done
