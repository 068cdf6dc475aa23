#!/bin/bash
# run bench with BBX_DEBUG variants, print device_ms_per_frame
for d in "$@"; do
  BBX_DEBUG=$d timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/dbg_$d.log 2>&1
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/dbg_$d.log").read().strip().splitlines()[-1])
    print("dbg=$d", round(d["value"],1), d["device_ms_per_frame_serial"])
except Exception as e:
    print("dbg=$d failed", e)
PY
done
