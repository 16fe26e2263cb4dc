"""Rows wider than 512 dims: the sliced MFMA path at a few shapes, and the one-thread-per-row kernel beside it."""
import sys, os
sys.path[:0] = ["/root/repo/tools"]
import quick_bench as qb
from vector_quantization import native
print(native.device_info())
for c in [(262144,1024,512),(262144,1024,1024),(65536,8192,1024),(131072,1024,768),(65536,1024,2048),(16384,16384,1024)]:
    qb.bench(*c)
qb.bench(8192,1024,1024)
qb.bench(8192,1024,1024, iters=2, flags=native.F_FORCE_SIMPLE)

# the similarity matrix at D = 1024 (65536 x 1024 entries): sliced MFMA sweep vs one thread per entry
import time
import torch
x = torch.randn(1, 65536, 1024, device="cuda:0")
cb = torch.randn(1, 1024, 1024, device="cuda:0")
out = torch.empty(1, 65536, 1024, device="cuda:0")
for name, fl, it in (("sliced sweep", 0, 10), ("one thread per entry", native.F_FORCE_SIMPLE, 2)):
    native.similarities(x, cb, flags=fl, out=out); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        native.similarities(x, cb, flags=fl, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / it
    print(f"similarities M=65536 K=1024 D=1024, {name}: {ms:.3f} ms ({2*65536*1024*1024/ms/1e9:.1f} TFLOP/s)", flush=True)
