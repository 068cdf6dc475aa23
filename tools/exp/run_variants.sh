#!/bin/bash
# GPU box: bench each variant library, print throughput + device_ms_per_frame
cp blackbox_amd/libbbx_hip.so /tmp/libbbx_orig.so
for f in tools/exp/variants/libbbx_*.so; do
  name=$(basename $f .so); name=${name#libbbx_}
  cp $f blackbox_amd/libbbx_hip.so
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/var_$name.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/var_$name.log").read().strip().splitlines()[-1])
    print("$name", round(d["value"],1), {k: round(v,4) for k,v in d["device_ms_per_frame_serial"].items()}, d["lacosmic_stats"][6:8])
except Exception as e:
    print("$name failed", e)
PY
done
cp /tmp/libbbx_orig.so blackbox_amd/libbbx_hip.so
