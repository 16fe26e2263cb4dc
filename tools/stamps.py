"""Diagnostic: per-wave phase durations (prologue / sweep / finalize) from in-kernel s_memtime stamps.

Build the diagnostic library first:  VQ_EXTRA_FLAGS=-DVQ_EXP_STAMPS ./build.sh && mv lib/libvq_mi355x.so lib/stamps.so
then run with VQ_MI355X_LIB=.../lib/stamps.so python tools/stamps.py M,K,D
"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch, numpy as np
from vector_quantization import native
M, K, D = (int(v) for v in sys.argv[1].split(","))
BWD = len(sys.argv) > 2 and sys.argv[2] == "bwd"  # stamps of vq_ce_backward (4-wave workgroups) instead of the search
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
x = torch.randn((1, M, D), generator=g).to(dev); cb = torch.randn((1, 1, K, D), generator=g).to(dev)
packed = native.pack_codebooks(cb, 0)
if BWD:
    tgt = torch.randint(0, K, (1, M), generator=g).to(dev)
    lse, tl = native.softmax_stats(x, cb[:, 0], target=tgt, packed=packed)
    coef = torch.tensor([1.0 / M], device=dev)
    for _ in range(3):
        native.ce_backward(x, cb[:, 0], lse, tl, tgt, coef, packed=packed)
else:
    for _ in range(3):
        native.quantize(x, cb, packed=packed, want_best=False)
torch.cuda.synchronize()
lib = native.load()
buf = (ctypes.c_uint64 * (8192 * 4))()
lib.vq_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.vq_debug_read_stamps(buf, 8192 * 4)
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
nw = min(8192, (M + 31) // 32)
st = st[:nw]
pro = st[:, 1] - st[:, 0]; sweep = st[:, 2] - st[:, 1]; fin = st[:, 3] - st[:, 2]
print("rc", rc, "waves", nw)
for name, a in (("prologue", pro), ("sweep", sweep), ("finalize", fin), ("total", st[:, 3] - st[:, 0])):
    print(f"{name:9s} cycles: median {np.median(a):10.0f}  p10 {np.percentile(a,10):10.0f}  p90 {np.percentile(a,90):10.0f}")
t0 = st[:, 0].min()
print("kernel span cycles (memtime ticks):", st[:, 3].max() - t0)
print("start-time spread of first-round waves:", np.percentile(st[:, 0] - t0, [0, 25, 50, 75, 100]))
