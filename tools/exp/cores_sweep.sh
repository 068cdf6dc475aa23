#!/bin/bash
# GPU box: headline rate of ONE rank on 16 / 8 / 4 cores (taskset), and two ranks sharing the GPU on 8 cores each
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4h/cores.log; mkdir -p gpurun_out/r4h; : > $out
run() {  # label, command...
  label=$1; shift
  "$@" > /tmp/c.json 2>/tmp/c.err
  python3 - "$label" >> $out <<PY
import json,sys
try:
    d=json.loads(open("/tmp/c.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps %.1f" % d["value"], "n_gpus", d["n_gpus"], "workers", d["config"]["host_fit_workers"], d["host_ms_per_frame"])
except Exception as e:
    print(sys.argv[1], "ERR", e, open("/tmp/c.err").read()[-400:])
PY
}
run "1 rank, 16 cores" timeout -k 10 250 taskset -c 0-15 python bench.py --steps 60 --warmup 6 --no-extras --no-cpu
run "1 rank, 8 cores" timeout -k 10 250 taskset -c 0-7 python bench.py --steps 60 --warmup 6 --no-extras --no-cpu
run "1 rank, 4 cores" timeout -k 10 250 taskset -c 0-3 python bench.py --steps 60 --warmup 6 --no-extras --no-cpu
BBX_BENCH_BACKEND=gloo BBX_BENCH_ONE_GPU=1 run "2 ranks on one GPU, 16 cores together" timeout -k 10 400 taskset -c 0-15 python bench.py --gpus 2 --steps 40 --warmup 6 --no-extras --no-cpu --lanes 4 --depth 10
cat $out
