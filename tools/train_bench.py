"""Training-side timings of the modules (cfg2 VectorQuantize, cfg4 ResidualVQ): forward with a frozen codebook, forward with
the EMA update, forward + backward.  python tools/train_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch
import vector_quantization as vq
from vector_quantization.codebooks import CodebookParams

dev = "cuda:0"


def timed(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


torch.manual_seed(0)
for name, mod, shape in (
        ("cfg2 VectorQuantize(256, K=1024) [256,1024,256]",
         vq.VectorQuantize(dim=256, codebook_params=CodebookParams(dim=256, codebook_size=1024, threshold_ema_dead_code=0)), (256, 1024, 256)),
        ("cfg4 ResidualVQ(256, Q=8, K=1024) [64,1024,256]",
         vq.ResidualVQ(dim=256, num_quantizers=8, codebook_params=CodebookParams(dim=256, codebook_size=1024, threshold_ema_dead_code=0)),
         (64, 1024, 256))):
    mod = mod.to(dev)
    x = torch.randn(shape, device=dev)
    xg = x.clone().requires_grad_(True)
    with torch.no_grad():
        mod.eval()
        t_eval = timed(lambda: mod(x))
        mod.train()
        t_frozen = timed(lambda: mod(x, freeze_codebook=True))
        t_ema = timed(lambda: mod(x))

    def step():
        q, _i, loss = mod(xg)
        (q.sum() * 1e-3 + loss.sum()).backward()
        xg.grad = None

    t_step = timed(step)
    print(f"{name}: eval {t_eval:.3f} ms | train forward, frozen codebook {t_frozen:.3f} | train forward + EMA update {t_ema:.3f} | "
          f"forward + EMA + backward {t_step:.3f}", flush=True)

# EMA accumulation alone at cfg2: float atomics vs the reproducible (atomics-free) variant
from vector_quantization import native

x = torch.randn(1, 262144, 256, device=dev)
idx = torch.randint(0, 1024, (1, 262144), device=dev)
t_a = timed(lambda: native.ema_accumulate(x, idx, 1024))
t_d = timed(lambda: native.ema_accumulate(x, idx, 1024, deterministic=True))
print(f"EMA accumulate, M=262144 K=1024 D=256 (incl. zero-fill of the outputs): atomic {t_a:.3f} ms | reproducible {t_d:.3f} ms", flush=True)
