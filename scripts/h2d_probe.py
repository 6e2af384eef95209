"""PCIe probe for the host-ABI analysis: pinned H2D / D2H rates of this box, alone and under a running kernel."""
import time
import torch

dev = torch.device("cuda", 0)
n = 256 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device=dev)
big = torch.randn(64 << 20, device=dev)
s_copy = torch.cuda.Stream()


def rate(fn, reps=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best / 1e9


def h2d():
    with torch.cuda.stream(s_copy):
        d.copy_(h, non_blocking=True)


def h2d_chunks(chunk):
    with torch.cuda.stream(s_copy):
        for o in range(0, n, chunk):
            d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)


def d2h():
    with torch.cuda.stream(s_copy):
        h.copy_(d, non_blocking=True)


def busy_h2d():
    for _ in range(40):
        big.mul_(1.0001)          # streams 512 MB per launch on the default stream while the copy runs
    h2d()


print("H2D one 256 MB copy      %.1f GB/s" % rate(h2d))
print("H2D 16 MB chunks         %.1f GB/s" % rate(lambda: h2d_chunks(16 << 20)))
print("H2D 32 MB chunks         %.1f GB/s" % rate(lambda: h2d_chunks(32 << 20)))
print("D2H one 256 MB copy      %.1f GB/s" % rate(d2h))
t0 = time.perf_counter(); busy_h2d(); torch.cuda.synchronize(); t = time.perf_counter() - t0
print("H2D under a streaming kernel: copy + 40 launches took %.2f ms" % (t * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): big.mul_(1.0001)
torch.cuda.synchronize(); print("the 40 launches alone %.2f ms" % ((time.perf_counter() - t0) * 1e3))
import os
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/class/drm/card0/device/numa_node").read().strip(), "numa node of card0")
except Exception as e:
    print("numa:", e)
