"""one-screen summary of a bench.py JSON line: python3 tools/dbg/bench_sum.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.1f %s  ms_per_step %.3f  idle_to_idle %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('idle_to_idle', {}).get('frames_per_s')))
r = d['roofline']
print('group ms live %.3f alone %.3f  frac %.4f alone %.4f  frac_moved %.3f  moved %.2f GB  traffic %s' % (
    r['avg_launch_ms'], r.get('alone', {}).get('avg_launch_ms', 0), r['frac'], r.get('alone', {}).get('frac', 0), r.get('frac_moved', 0),
    r.get('moved_bytes_by_design', 0) / 1e9, r.get('traffic')))
for k, v in r.get('kernels', {}).items():
    print('  %-14s live %.3f alone %.3f  moved %.2f GB  frac_moved %.2f' % (k, v['avg_launch_ms'], v.get('alone_ms', 0), v['moved_bytes_per_launch'] / 1e9, v['frac_moved']))
for k, v in r.get('others', {}).items():
    print('  %-14s live %.3f  frac_moved %.2f' % (k, v['avg_launch_ms'], v['frac_moved']))
print('frame.frac %.3f' % r['frame']['frac'])
for k in ('single_frame_latency_ms', 'long_run', 'host_ms_per_frame'):
    if k in d: print(k, d[k])
