#!/bin/bash
# GPU box: bench.py's CLI list run (child `python blackbox.py --image_list`, 96 files) under different environments
#   usage: tools/exp/cli_sweep.sh "VAR=val VAR2=val" "VAR=val" ...      -> gpurun_out/cli_sweep.txt
cd "$GRAFT_REPO_ROOT" || exit 1
: > gpurun_out/cli_sweep.txt
for e in "$@"; do
  env BBX_CLI_NO_POOL=1 $e timeout -k 10 400 python3 bench.py --proc-only > gpurun_out/cli_sweep.json 2> gpurun_out/cli_sweep.err || { echo "$e: failed" >> gpurun_out/cli_sweep.txt; tail -3 gpurun_out/cli_sweep.err >> gpurun_out/cli_sweep.txt; continue; }
  python3 - "$e" <<'PY' >> gpurun_out/cli_sweep.txt
import json, sys
d = json.loads(open('gpurun_out/cli_sweep.json').read().strip().splitlines()[-1])['image_list']['ramdisk']
print('%-40s steady %.1f  whole list %.1f  process %.1f frames/s  (products of %s files)' % (sys.argv[1], d.get('frames_per_s', 0), d.get('frames_per_s_whole_list', 0), d.get('frames_per_s_process', 0), d.get('products_of')))
print('      ', d.get('pipeline'), 'first product after', d.get('seconds_to_first_product'), 's; file 16 after', d.get('seconds_to_product_16'), 's; before the list', d.get('seconds_before_the_list'), d.get('cgroup_cpu'))
PY
done
cat gpurun_out/cli_sweep.txt
