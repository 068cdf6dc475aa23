#!/bin/bash
# GPU box: per kernel of serial frames: vector instructions per launch, duration, and the share of the launch's SIMD issue
# slots they fill (instructions x 4 cycles / (1024 SIMDs x duration x 2.1 GHz)) -> gpurun_out/valu_frac.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/vf
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/vf -o r -- python3 tools/prof_serial.py 3 > gpurun_out/vf.log 2>&1 || { tail -5 gpurun_out/vf.log; exit 1; }
python3 - <<'PY' > gpurun_out/valu_frac.txt
import csv, glob, collections
cc = (glob.glob('gpurun_out/vf/*counter_collection.csv') + glob.glob('gpurun_out/vf/*/*counter_collection.csv'))[0]
kt = (glob.glob('gpurun_out/vf/*kernel_trace.csv') + glob.glob('gpurun_out/vf/*/*kernel_trace.csv'))[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); tdur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(cc)):
    d = r['Dispatch_Id']
    if d not in dur: continue
    name = dur[d][1].split('(')[0].replace('void ', '')[:56]
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    if d not in seen:
        seen.add(d); n[name] += 1; tdur[name] += dur[d][0]
rows = []
for name in acc:
    v = acc[name]['SQ_INSTS_VALU'] / n[name]; us = tdur[name] / n[name] / 1e3
    frac = v * 4 / (1024 * us * 1e-6 * 2.1e9) if us > 0 else 0
    rows.append((tdur[name], name, n[name], us, v / 1e6, acc[name]['SQ_INSTS_SALU'] / n[name] / 1e6, acc[name]['SQ_INSTS_LDS'] / n[name] / 1e6, frac))
print('%-56s %5s %9s %9s %9s %9s %6s' % ('kernel (counters serialise kernels: durations are longer than in a plain run)', 'calls', 'us', 'VALU M', 'SALU M', 'LDS M', 'issue'))
for t, name, k, us, v, sa, l, f in sorted(rows, reverse=True)[:40]:
    print('%-56s %5d %9.1f %9.2f %9.2f %9.2f %6.2f' % (name, k, us, v, sa, l, f))
PY
rm -rf gpurun_out/vf
cat gpurun_out/valu_frac.txt
