#!/bin/bash
# GPU box: repeated headline runs: rate and the ZOGY kernels' launch times, to see what differs between fast and slow runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; : > gpurun_out/modes.txt
for r in $(seq ${1:-8}); do
  timeout -k 10 150 python3 bench.py --no-cpu --no-extras --steps 300 --warmup 20 ${BENCH_ARGS} > gpurun_out/modes_one.json 2> gpurun_out/modes_one.err || { tail -5 gpurun_out/modes_one.err; exit 1; }
  python3 -c "
import json
for l in open('gpurun_out/modes_one.json'):
    if l.startswith('{'):
        d = json.loads(l); k = d['roofline']['kernels']; o = d['roofline']['others']
        print('%.1f fps | zogy group %.2f ms |' % (d['value'], d['roofline']['avg_launch_ms']), ' '.join('%s %.2f' % (n[2:], v['avg_launch_ms']) for n, v in k.items()), '|', ' '.join('%s %.2f' % (n[2:], v['avg_launch_ms']) for n, v in o.items()), '| lat %.1f' % d['single_frame_latency_ms'])
" | tee -a gpurun_out/modes.txt
done
