#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4h/calib.log; mkdir -p gpurun_out/r4h; : > $out
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload calib --steps 300 --warmup 30 --no-extras --no-cpu $EXTRA > /tmp/c.json 2>/tmp/c.err
  python3 - "$label" >> $out <<PY
import json,sys
try:
    d=json.loads(open("/tmp/c.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps %.0f" % d["value"], d["host_ms_per_frame"], {k: round(v,2) for k,v in d["pipeline_wall_ms_per_frame"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/c.err").read()[-300:])
PY
}
EXTRA="" run "chunk4" BBX_HOST_CHUNK=4
EXTRA="" run "chunk8" BBX_HOST_CHUNK=8
EXTRA="" run "chunk16" BBX_HOST_CHUNK=16
EXTRA="--depth 40" run "chunk16 depth40" BBX_HOST_CHUNK=16
EXTRA="--depth 40 --workers 14" run "chunk16 depth40 w14" BBX_HOST_CHUNK=16
EXTRA="" run "chunk4 again" BBX_HOST_CHUNK=4
cat $out
