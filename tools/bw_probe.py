#!/usr/bin/env python
"""HBM bandwidth reference points on this box: torch copy / read-only reduce / write-only fill on 1 GiB."""
import torch, time
n = 256 * 1024 * 1024
x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: y.copy_(x)); print("copy   1 GiB->1 GiB: %.3f ms  %.2f TB/s (read+write)" % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: x.sum());    print("reduce 1 GiB       : %.3f ms  %.2f TB/s (read)" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: y.fill_(1.)); print("fill   1 GiB       : %.3f ms  %.2f TB/s (write)" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: torch.add(x, y, out=y)); print("add    2 GiB->1 GiB: %.3f ms  %.2f TB/s" % (ms, 3 * n * 4 / ms / 1e9))
