#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/d2h; : > gpurun_out/d2h/log.txt
i=0
for cfg in "A=1" "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=0"; do
  i=$((i+1)); rm -rf gpurun_out/d2h/t$i
  env $cfg timeout -k 10 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/d2h/t$i -o r -- python3 tools/exp/d2h_path.py > gpurun_out/d2h/o$i.txt 2>&1
  echo "== $cfg: $(grep 'GB/s' gpurun_out/d2h/o$i.txt)" >> gpurun_out/d2h/log.txt
  f=$(ls gpurun_out/d2h/t$i/*/*kernel_stats.csv gpurun_out/d2h/t$i/*kernel_stats.csv 2>/dev/null | head -1)
  grep -i "copyBuffer" $f | cut -c1-160 >> gpurun_out/d2h/log.txt
  m=$(ls gpurun_out/d2h/t$i/*/*memory_copy_stats.csv gpurun_out/d2h/t$i/*memory_copy_stats.csv 2>/dev/null | head -1)
  [ -n "$m" ] && cat $m | cut -c1-200 >> gpurun_out/d2h/log.txt
  rm -rf gpurun_out/d2h/t$i
done
cat gpurun_out/d2h/log.txt
