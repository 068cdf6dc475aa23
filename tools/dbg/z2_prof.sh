#!/bin/bash
# GPU box: per-kernel times of bbx_zogy_frame alone (rocprofv3 kernel stats of tools/dbg/z2_time.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/z2p
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/z2p -o r -- python3 tools/dbg/z2_time.py > gpurun_out/z2p.log 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/z2p 7 50 | grep "z2::\|total"
rm -rf gpurun_out/z2p
