#!/bin/bash
# GPU box: the launch sequence of ONE serial frame (name, start offset, duration, gap to the previous end, grid) -> gpurun_out/frame_seq.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/fseq
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fseq -o r -- python3 tools/prof_serial.py 4 > gpurun_out/fseq.log 2>&1 || exit 1
python3 - <<'PY' > gpurun_out/frame_seq.txt
import csv
rows = list(csv.DictReader(open('gpurun_out/fseq/r_kernel_trace.csv')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']), int(r['Workgroup_Size_X'])) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].startswith('void k_calibrate_v4')]
a, b = starts[-2], starts[-1]
t0 = ev[a][0]; prev = t0; busy = 0
for s, e, name, grid, wg in ev[a:b]:
    short = name.split('(')[0].replace('void ', '')[:60]
    print('%9.1f  dur %8.1f  gap %7.1f  grid %9d x%4d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, grid // max(wg, 1), wg, short))
    prev = max(prev, e); busy += (e - s)
print('frame %.1f us, kernels %.1f us, launches %d' % ((ev[b][0] - t0) / 1e3, busy / 1e3, b - a))
PY
rm -rf gpurun_out/fseq
tail -3 gpurun_out/frame_seq.txt
