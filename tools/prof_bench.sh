#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run, per-frame summary -> gpurun_out/<tag>.txt
# usage: tools/prof_bench.sh <tag> [steps]
tag=${1:-prof}; steps=${2:-30}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o r -- python3 bench.py --steps $steps --warmup 4 --no-cpu $BENCH_EXTRA > gpurun_out/$tag.log 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/$tag $((steps + 16)) > gpurun_out/$tag.txt
rm -f gpurun_out/$tag/r_results.db gpurun_out/$tag/r_kernel_trace.csv
head -${3:-16} gpurun_out/$tag.txt; tail -1 gpurun_out/$tag.txt
