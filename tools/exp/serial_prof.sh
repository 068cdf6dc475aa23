#!/bin/bash
# GPU box: kernels of serial frames alone on the GPU -> gpurun_out/<tag>_serial.txt
tag=${1:-ps}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r -- python3 tools/prof_serial.py 9 > gpurun_out/$tag.log 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/$tag 10 3 > gpurun_out/${tag}_serial.txt
rm -rf gpurun_out/$tag
head -70 gpurun_out/${tag}_serial.txt
