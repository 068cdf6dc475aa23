#!/bin/bash
# GPU box: the fpack kernels alone -> gpurun_out/fpack_kernel_summary.txt (copy to profiles/rNN_fpack_kernel_summary.txt)
#   1. rocprofv3 --kernel-trace --stats of tools/dbg/fp_time.py in the product configuration (FP_ONLY=1: float image at
#      q = 16 / 4 / 2 and the uint8 mask, 12 launches each of a 10560 x 10560 image): average duration per kernel
#   2. SQ counters of the same program (tools/dbg/fp_pmc.sh: separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export FP_ONLY=1
OUT=gpurun_out/fp_prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python3 tools/dbg/fp_time.py > $OUT/time.log 2>&1 || { tail -5 $OUT/time.log; exit 1; }
{
  echo "# tools/dbg/fp_prof.sh at $(date -u +%Y-%m-%dT%H:%MZ): bbx_fpack_tiles on a 10560 x 10560 image, product configuration"
  echo "# (two workgroups per CU, bracket medians with row hints); event-timed calls (both launches of a call):"
  grep "hist_only" $OUT/time.log
  echo
  echo "# rocprofv3 --kernel-trace --stats: per kernel"
  python3 - $OUT/kt <<'PY'
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*kernel_stats.csv') + glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'))[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_fp_' in n: print('%-70s calls %5s  avg %9.1f us  min %9.1f  max %9.1f' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  echo
  echo "# SQ counters per launch (averages over the launches of each instantiation; float: q = 16 / 4 / 2 together)"
} > gpurun_out/fpack_kernel_summary.txt
bash tools/dbg/fp_pmc.sh > /dev/null 2>&1
grep -A16 "k_fp_tile" gpurun_out/fp_pmc.txt >> gpurun_out/fpack_kernel_summary.txt
rm -rf $OUT
cat gpurun_out/fpack_kernel_summary.txt
