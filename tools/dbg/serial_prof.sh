#!/bin/bash
# GPU box: rocprofv3 kernel statistics of serial frames (kernels alone on the GPU) -> gpurun_out/serial_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/ps; rm -rf $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python3 tools/prof_serial.py 9 > gpurun_out/ps.log 2>&1 || { tail -5 gpurun_out/ps.log; exit 1; }
python3 tools/prof_summary.py $OUT 10 0.5 > gpurun_out/serial_summary.txt
awk -F'calls/frame=' '/calls\/frame/{split($2,a," "); s+=a[1]} END{print "launches/frame", s}' gpurun_out/serial_summary.txt
tail -3 gpurun_out/serial_summary.txt | cut -c1-200
rm -rf $OUT
