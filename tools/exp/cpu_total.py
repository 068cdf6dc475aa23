"""GPU box: CPU seconds (user + system) of the whole bench process tree per wall second, sampled while it runs"""
import subprocess, sys, time, os
import psutil
p = subprocess.Popen([sys.executable, 'bench.py'] + (sys.argv[1:] or ['--no-cpu', '--no-extras', '--steps', '600', '--warmup', '20']), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
proc = psutil.Process(p.pid)
samples = []
t0 = time.time()
while p.poll() is None:
    time.sleep(0.5)
    try:
        procs = [proc] + proc.children(recursive=True)
        tot = 0.0; per = []
        for q in procs:
            try:
                c = q.cpu_times(); tot += c.user + c.system; per.append((q.pid, c.user + c.system, q.num_threads()))
            except psutil.Error:
                pass
        samples.append((time.time() - t0, tot, len(procs), per))
    except psutil.Error:
        pass
out = p.stdout.read().decode()
for i in range(1, len(samples)):
    dt = samples[i][0] - samples[i - 1][0]
    print('t=%5.1f s  cores busy %.2f  (%d processes)' % (samples[i][0], (samples[i][1] - samples[i - 1][1]) / dt, samples[i][2]))
import json
for l in out.splitlines():
    if l.startswith('{'):
        print('fps', json.loads(l)['value'])
