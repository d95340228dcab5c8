"""Pageable host memory <-> device, one copy against slices copied by several host threads at once."""
import sys, time, threading
import numpy as np, torch
n = 256 << 20
dev = torch.device('cuda', 0)
d = torch.empty(n, dtype=torch.uint8, device=dev)
h = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
ht = torch.from_numpy(h)
back = torch.empty(n, dtype=torch.uint8)
streams = [torch.cuda.Stream(dev) for _ in range(8)]

def run(T, direction):
    step = (n + T - 1) // T
    def work(k):
        lo, hi = k * step, min(n, (k + 1) * step)
        with torch.cuda.stream(streams[k]):
            if direction == 'h2d':
                d[lo:hi].copy_(ht[lo:hi], non_blocking=False)
            else:
                back[lo:hi].copy_(d[lo:hi], non_blocking=False)
            streams[k].synchronize()
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        best = min(best, time.perf_counter() - t0)
    return best

for direction in ('h2d', 'd2h'):
    for T in (1, 2, 4, 8):
        t = run(T, direction)
        print(f'{direction} {T} thread(s): {t * 1e3:.1f} ms, {n / 1e9 / t:.1f} GB/s', flush=True)
assert torch.equal(back, ht)
# both directions at once, 4 threads each
def both():
    a = threading.Thread(target=lambda: run(4, 'h2d')); b = threading.Thread(target=lambda: run(4, 'd2h'))
    t0 = time.perf_counter(); a.start(); b.start(); a.join(); b.join(); return (time.perf_counter() - t0) / 4
print(f'both directions, 4 threads each: {both() * 1e3:.1f} ms per pair of 256 MiB copies')
