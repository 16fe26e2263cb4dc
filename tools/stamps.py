"""Diagnostic: per-wave phase durations (prologue / sweep / finalize) from in-kernel s_memtime stamps.

Build the diagnostic library first:  VQ_EXTRA_FLAGS=-DVQ_EXP_STAMPS ./build.sh && mv lib/libvq_mi355x.so lib/stamps.so
then run with VQ_MI355X_LIB=.../lib/stamps.so python tools/stamps.py M,K,D
"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch, numpy as np
from vector_quantization import native
vals = [int(v) for v in sys.argv[1].split(",")]
M, K, D = vals[:3]
Q = vals[3] if len(vals) > 3 else 1
BWD = len(sys.argv) > 2 and sys.argv[2] == "bwd"  # stamps of vq_ce_backward (4-wave workgroups) instead of the search
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
x = torch.randn((1, M, D), generator=g).to(dev); cb = torch.randn((1, Q, K, D), generator=g).to(dev)
packed = native.pack_codebooks(cb, 0)
if BWD:
    tgt = torch.randint(0, K, (1, M), generator=g).to(dev)
    lse, tl = native.softmax_stats(x, cb[:, 0], target=tgt, packed=packed)
    coef = torch.tensor([1.0 / M], device=dev)
    for _ in range(3):
        native.ce_backward(x, cb[:, 0], lse, tl, tgt, coef, packed=packed)
else:
    import time
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.1:  # steady state: the clock settles after ~30 ms of load
        for _ in range(5):
            native.quantize(x, cb, packed=packed, want_best=False)
        torch.cuda.synchronize()
torch.cuda.synchronize()
lib = native.load()
NST = 64
buf = (ctypes.c_uint64 * (8192 * NST))()
lib.vq_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.vq_debug_read_stamps(buf, 8192 * NST)
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, NST).astype(np.int64)
nw = min(8192, (M + 15) // 16 if (D > 256 and Q == 1) else (M + 31) // 32)  # wave pairs: 8 waves per 128 rows
st = st[:nw]
pro = st[:, 1] - st[:, 0]; sweep = st[:, 2] - st[:, 1]; fin = st[:, 3] - st[:, 2]
print("rc", rc, "waves", nw)
for name, a in (("prologue", pro), ("sweep", sweep), ("finalize", fin), ("total", st[:, 3] - st[:, 0])):
    print(f"{name:9s} cycles: median {np.median(a):10.0f}  p10 {np.percentile(a,10):10.0f}  p90 {np.percentile(a,90):10.0f}")
t0 = st[:, 0].min()
print("kernel span cycles (memtime ticks):", st[:, 3].max() - t0)
print("start-time spread of first-round waves:", np.percentile(st[:, 0] - t0, [0, 25, 50, 75, 100]))

if not BWD:
    # per-stage stamps: 4 + 5 q + {3: stage top, 0: sweep start (norm done), 1: sweep done, 2: merged + idx stored, 4: residual updated}
    print("stage:   norm   sweep  resolve  update   (median cycles per wave)")
    for q in range(Q):
        b = 4 + 5 * q
        if b + 4 >= NST:
            break
        seg = [st[:, b + 0] - st[:, b + 3], st[:, b + 1] - st[:, b + 0], st[:, b + 2] - st[:, b + 1], st[:, b + 4] - st[:, b + 2]]
        print(f"  q={q}: " + "  ".join(f"{np.median(a):8.0f}" for a in seg))

rt = (st[:, 61] - st[:, 60]).astype(np.float64)  # 100 MHz ticks
cyc = (st[:, 3] - st[:, 0]).astype(np.float64)
ok = rt > 0
print(f"shader clock over a wave's life: median {np.median(cyc[ok] / rt[ok]) * 100:.0f} MHz  (p10 {np.percentile(cyc[ok] / rt[ok], 10) * 100:.0f}, p90 {np.percentile(cyc[ok] / rt[ok], 90) * 100:.0f})")
print(f"wave life: median {np.median(rt[ok]) / 100:.1f} us; first start .. last end (100 MHz counter): {(st[:, 61].max() - st[:, 60].min()) / 100:.1f} us")
