#!/bin/bash
# GPU box: headline and files-to-files rate with PyTorch on its bundled HIP runtime (ROCm 7.0: D2H copies are blit kernels) and on
# the system's (ROCm 7.2: D2H copies go to the copy engines), A / B / A / B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4k; out=gpurun_out/r4k/sysrt.log; : > $out
PRE=/opt/rocm/lib/libamdhip64.so.7:/opt/rocm/lib/libhsa-runtime64.so.1
for rep in 1 2; do
for v in bundled system; do
  if [ $v = system ]; then e=(LD_PRELOAD=$PRE); else e=(A=1); fi
  env "${e[@]}" timeout -k 10 300 python bench.py --io-only --io-simple --steps 60 > /tmp/io.json 2>/tmp/io.err
  env "${e[@]}" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-extras --no-cpu > /tmp/b.json 2>/tmp/b.err
  python3 - "$v" >> $out <<PY
import json,sys
try:
    r=json.loads(open("/tmp/io.json").read())["ramdisk"]
    io="io fps %.1f %s" % (r["frames_per_s"], r.get("error",""))
except Exception as e:
    io="io ERR %s %s" % (e, open("/tmp/io.err").read()[-300:])
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    hl="headline %.1f latency %.1f" % (d["value"], d["single_frame_latency_ms"])
except Exception as e:
    hl="headline ERR %s %s" % (e, open("/tmp/b.err").read()[-300:])
print(sys.argv[1], io, hl)
PY
done; done
cat $out
