#!/usr/bin/env python3
"""What does HBM take when it is mostly WRITTEN?  (development aid, GPU only)  The assemble kernels write three bytes for every
byte they read (9 B per merged base out, 2 B per read base in); the 8 TB/s of the roofline is the pins' rate either way, what a
stream of stores sustains is measured here: fill (write only), copy (one read per write), and a 1 : 3 read : write mix."""
import time
import torch

n = 1 << 32     # bytes
x = torch.empty(n // 8, dtype=torch.int64, device="cuda:0")
y = torch.empty(n // 8, dtype=torch.int64, device="cuda:0")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: x.fill_(7))
print("fill   (write only)        %6.2f TB/s of writes" % (n / t / 1e12))
t = timed(lambda: y.copy_(x))
print("copy   (1 read : 1 write)  %6.2f TB/s total (%.2f written)" % (2 * n / t / 1e12, n / t / 1e12))
q = n // 8 // 4
def mix():
    torch.add(x[:q], 1, out=y[:q])
    y[q:].fill_(3)
t = timed(mix)
print("mix    (1 read : 4 writes, two kernels back to back) %6.2f TB/s total (%.2f written)" % ((n / 4 + n) / t / 1e12, n / t / 1e12))
# the same mix inside ONE kernel: out[i] = in[i // 4] (every input word read once, four words written)
z = torch.empty(n // 8 // 4, dtype=torch.int64, device="cuda:0")
t = timed(lambda: torch.repeat_interleave(z, 4, output_size=n // 8))
print("mix    (1 read : 4 writes, one kernel: repeat_interleave, allocates its output) %6.2f TB/s total" % ((n / 4 + n) / t / 1e12))
t = timed(lambda: x.sum())
print("sum    (read only)         %6.2f TB/s of reads" % (n / t / 1e12))
