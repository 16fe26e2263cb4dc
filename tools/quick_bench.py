"""Quick kernel timing (dev tool): python tools/quick_bench.py [M,K,D[,H[,Q[,ste]]]] ...
Each case: ~80 ms of untimed load first (the chip's clock needs ~30 ms to settle), then 20 timed launches."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd"), os.path.join(ROOT, "tests", "golden")]
import torch
from vector_quantization import native


def bench(M, K, D, H=1, Q=1, ste=0, iters=20, flags=0, want_out=True):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    x = torch.randn((H, M, D), generator=g).to(dev)
    cb = torch.stack([torch.randn((H, K, D), generator=g) * 2.0 ** (-i / 2.0) for i in range(Q)], dim=1).contiguous().to(dev)
    packed = native.pack_codebooks(cb, 0)
    kw = dict(packed=packed, flags=flags, want_out=want_out, want_best=False, ste=bool(ste), want_sq_err=bool(ste))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:
        native.quantize(x, cb, **kw)
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        native.quantize(x, cb, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    flops = 2.0 * H * M * K * D * Q
    print(f"M={M} K={K} D={D} H={H} Q={Q} ste={ste}: {ms:.4f} ms  {H*M/ms/1e3:.2f} Mrows/s  {flops/ms/1e9:.1f} TFLOP/s ({flops/ms/1e9/157.3*100:.1f}% of 157.3)", flush=True)


if __name__ == "__main__":
    print(native.device_info())
    cfgs = [(262144, 1024, 256), (262144, 8192, 256), (65536, 8192, 64, 8), (65536, 1024, 256, 1, 8), (65536, 8192, 512, 8), (8192, 65536, 512)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for c in cfgs:
        bench(*c)
