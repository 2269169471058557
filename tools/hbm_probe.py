"""Calibration: what torch's own fill / copy kernels reach on this GPU (achievable HBM bandwidth beside the 8 TB/s spec).
Measured on the MI355X box in round 1: fill 6.7 TB/s, copy 4.8 TB/s (read+write)."""
import torch, time
x = torch.empty(1 << 30, dtype=torch.int32, device="cuda")   # 4 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms = t(lambda: x.zero_()); print("fill  4GiB: %.3f ms  %.0f GB/s" % (ms, 4*1.0737e9/ms/1e6))
ms = t(lambda: y.copy_(x)); print("copy  4GiB: %.3f ms  %.0f GB/s (r+w)" % (ms, 8*1.0737e9/ms/1e6))
ms = t(lambda: x.sum()); print("read  4GiB: %.3f ms  %.0f GB/s" % (ms, 4*1.0737e9/ms/1e6))
