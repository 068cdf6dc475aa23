"""GPU box: what the host side of the output stage can move -- pinned D2H / H2D bandwidth with 1..8 copy streams, file
write rate of 1..8 threads to /dev/shm and the scratch directory (100 MB files)"""
import os, sys, time, threading, tempfile
import torch
dev = torch.device('cuda', 0)
N = 100 << 20
d = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(8)]
h = [torch.empty(N, dtype=torch.uint8, pin_memory=True) for _ in range(8)]
for k in (1, 2, 4, 8):
    st = [torch.cuda.Stream() for _ in range(k)]
    for direction in ('d2h', 'h2d', 'both'):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            for i in range(k):
                with torch.cuda.stream(st[i]):
                    if direction in ('d2h', 'both'):
                        h[i].copy_(d[i], non_blocking=True)
                    if direction == 'h2d' or (direction == 'both' and i % 2 == 0):
                        d[(i + 4) % 8].copy_(h[(i + 4) % 8], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nb = reps * k * N * (1.5 if direction == 'both' else 1)
        print('%d streams %-5s %.1f GB/s' % (k, direction, nb / dt / 1e9))
buf = h[0].numpy()
for root in ('/dev/shm', tempfile.gettempdir()):
    for k in (1, 2, 4, 8):
        def w(i):
            for r in range(4):
                p = os.path.join(root, 'bbx_bw_%d_%d' % (os.getpid(), i))
                with open(p, 'wb') as f:
                    f.write(memoryview(buf))
                os.unlink(p)
        ts = [threading.Thread(target=w, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        dt = time.perf_counter() - t0
        print('%s: %d writer threads %.1f GB/s' % (root, k, 4 * k * N / dt / 1e9))
print('cpus', len(os.sched_getaffinity(0)), open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '')
