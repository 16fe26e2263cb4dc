"""Diagnostic: per-wave phase timestamps (needs lib/stamps.so built with -DVQ_EXP_STAMPS)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch, numpy as np
from vector_quantization import native
M, K, D = (int(v) for v in sys.argv[1].split(","))
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
x = torch.randn((1, M, D), generator=g).to(dev); cb = torch.randn((1, 1, K, D), generator=g).to(dev)
packed = native.pack_codebooks(cb, 0)
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
buf2 = (ctypes.c_uint64 * (8192 * 8))()
lib.vq_debug_read_segs.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.vq_debug_read_segs(buf2, 8192 * 8)
sg = np.frombuffer(buf2, dtype=np.uint64).reshape(8192, 8).astype(np.float64)[:nw]
ntile = (K + 31) // 32
names = ["(7->0) loop/prefetch addr", "mfma g0 (4)", "epilogue(prev)", "mfma g1-2 (8)", "stage issue", "mfma g3.. (116)", "aug mfma", "barrier wait"]
tot = sg[:, 1:].sum(1).mean() + sg[:, 0].mean()
for i, nme in enumerate(names):
    print(f"  seg{i} {nme:28s}: {sg[:, i].mean() / ntile:9.0f} cycles/tile  ({100 * sg[:, i].mean() / tot:5.1f} %)")
print("  sum per tile:", tot / ntile)
