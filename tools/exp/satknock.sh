#!/bin/bash
# GPU box: headline rate with the satellite stage cut short after hist (5) / gauss (4) / canny_tile (3) / cc filter (2) / Hough + walk (1)
# (scratch library tools/exp/_var/satv, built by tools/exp/libvar.sh satv -DSATV bbx_canny.hip bbx_sat.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/satknock.txt
for lv in "$@"; do
  BBX_LIB_PATH=tools/exp/_var/satv/libbbx_hip.so BBX_DBG_SAT=$lv timeout -k 10 150 python3 tools/exp/knock.py none --steps 200 --warmup 20 > gpurun_out/knock_one.json 2> gpurun_out/knock_one.err || { tail -5 gpurun_out/knock_one.err; exit 1; }
  python3 -c "
import json
for l in open('gpurun_out/knock_one.json'):
    if l.startswith('{'):
        d = json.loads(l); print('BBX_DBG_SAT=%s  %.1f frames/s' % ('$lv', d['value']))
" | tee -a gpurun_out/satknock.txt
done
