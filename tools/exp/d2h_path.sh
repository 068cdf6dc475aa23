#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/d2h; : > gpurun_out/d2h/log.txt
for m in hiphost torch; do
  rm -rf gpurun_out/d2h/t_$m
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d2h/t_$m -o r -- python3 tools/exp/d2h_path.py $m > gpurun_out/d2h/o_$m.txt 2>&1
  echo "== $(grep -E 'GB/s|HSA_' gpurun_out/d2h/o_$m.txt)" >> gpurun_out/d2h/log.txt
  f=$(ls gpurun_out/d2h/t_$m/*/*kernel_stats.csv gpurun_out/d2h/t_$m/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -i "copyBuffer" $f | cut -c1-90 >> gpurun_out/d2h/log.txt
  rm -rf gpurun_out/d2h/t_$m
done
cat gpurun_out/d2h/log.txt
