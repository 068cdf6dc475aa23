#!/bin/bash
# GPU box: the round's profile evidence -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/ afterwards)
#   tools/profile_round.sh r03 <commit>
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 40 --warmup 4 --no-cpu --no-extras` -> kernel stats CSV + summary
# 2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; each alone with --kernel-trace) of a short single-lane run -> traffic JSON
# 3. serial frames (kernels alone on the GPU) -> summary
TAG=${1:-rXX}; COMMIT=${2:-unknown}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python3 bench.py --steps 40 --warmup 4 --no-cpu --no-extras > $OUT/bench_line.json 2> $OUT/kt.log || exit 1
# frames behind the statistics: depth + warm-up + steps + depth of the timed run, lanes warm-up frames, 5 serial frames
python3 tools/prof_summary.py $OUT/kt 87 3 > $OUT/kernel_summary.txt
cp $(ls $OUT/kt/*/*kernel_stats.csv $OUT/kt/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras --lanes 1 --depth 2 > $OUT/pmc_$c.log 2>&1 || exit 1
done
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $COMMIT > $OUT/pmc_traffic.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ps -o r -- python3 tools/prof_serial.py 9 > $OUT/ps.log 2>&1 || exit 1
python3 tools/prof_summary.py $OUT/ps 10 3 > $OUT/serial_kernel_summary.txt
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/ps
ls -la $OUT
