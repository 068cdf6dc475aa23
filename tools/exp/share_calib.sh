#!/bin/bash
# GPU box: tools/trace_share.py on the calib workload -> gpurun_out/share_calib.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/shr
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shr -o r -- python3 bench.py --workload ${WL:-calib} --steps 400 --warmup 40 --no-cpu --no-extras > gpurun_out/shr.json 2> gpurun_out/shr.err || exit 1
python3 tools/trace_share.py gpurun_out/shr k_calibrate_v4 > gpurun_out/share_${WL:-calib}.txt
rm -rf gpurun_out/shr
cat gpurun_out/share_${WL:-calib}.txt
