#!/bin/bash
# GPU box: tools/trace_share.py on a kernel trace of the headline pipeline -> gpurun_out/share.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/shr
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shr -o r -- python3 bench.py --steps 120 --warmup 10 --no-cpu --no-extras ${BENCH_ARGS} > gpurun_out/shr.json 2> gpurun_out/shr.err || exit 1
python3 tools/trace_share.py gpurun_out/shr > gpurun_out/share.txt
rm -rf gpurun_out/shr
cat gpurun_out/share.txt
