#!/bin/bash
# GPU box: the whole GPU suite, smoke, stage bench, default bench line, kernel statistics
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1; rc=$?; tail -3 gpurun_out/final_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 || exit 1
timeout -k 10 500 python3 tools/stage_bench.py --reps 5 > gpurun_out/stage_bench.json 2> gpurun_out/stage_bench.err || exit 1
timeout -k 10 600 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err || exit 1
cut -c1-400 gpurun_out/bench_line.json
bash tools/prof_bench.sh r01f 240 240 > gpurun_out/r01f_summary.txt || exit 1
tail -1 gpurun_out/r01f_summary.txt
