"""Which torch pool streams share a hardware queue?  HIP multiplexes its streams over a few HSA queues
(GPU_MAX_HW_QUEUES, 4 by default); two streams on one queue execute in order, so launches issued on them cannot
overlap.  Probe: a chain of one-thread spin kernels on each of two streams -- distinct queues run the two chains
side by side (time ~ one chain), a shared queue runs them one after the other (time ~ two chains)."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream(dev) for _ in range(n)]
CH = 4


def chain(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(CH):
            torch.cuda._sleep(400_000)          # a one-thread spin kernel: device-bound, leaves the chip empty


def timed(idx):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx:
        chain(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for i in range(n):
    timed([i])
one = min(timed([0]) for _ in range(3))
print(f"one chain of {CH} spin kernels: {one:.2f} ms")
cls = list(range(n))
for i in range(n):
    row = []
    for j in range(n):
        if i == j:
            row.append("  -  ")
            continue
        t = min(timed([i, j]) for _ in range(2))
        shared = t > 1.6 * one
        row.append(f"{t / one:4.2f}{'*' if shared else ' '}")
        if shared and j > i:
            cls[j] = cls[i]
    print(f"stream {i} (0x{streams[i].cuda_stream:x}):", " ".join(row))
print("queue classes (streams with the same number share a hardware queue):", cls)
