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
