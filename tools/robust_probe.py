import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd"), os.path.join(ROOT, "tests", "golden")]
import torch, numpy as np
from vector_quantization import native
from oracle import vq_oracle as o
dev = torch.device("cuda:0")
def check(H, M, K, D, sample=512, metric=0):
    g = torch.Generator().manual_seed(H * 7 + K)
    x = torch.randn((H, M, D), generator=g)
    cb = torch.randn((H, 1, K, D), generator=g)
    r = native.quantize(x.to(dev), cb.to(dev), metric=metric, want_sq_err=True)
    torch.cuda.synchronize()
    rows = torch.randperm(M, generator=g)[:sample]
    rows = torch.cat([rows, torch.tensor([0, M - 1])])
    ok = True
    for h in range(min(H, 3)):
        ri, rb = o.nearest(x[h, rows].numpy(), cb[h, 0].numpy(), metric)
        ok &= bool((r["idx"][h, rows, 0].cpu().numpy() == ri).all())
        ok &= bool(np.array_equal(r["best"][h, rows, 0].cpu().numpy().view(np.uint32), rb.view(np.uint32)))
    hh = torch.arange(H, device=dev)[:, None]
    ok &= bool(torch.equal(r["out"], cb.to(dev)[:, 0][hh, r["idx"][..., 0]]))
    print(f"H={H} M={M} K={K} D={D} metric={metric}: {'OK' if ok else 'MISMATCH'}")
    del r
    torch.cuda.empty_cache()
check(1, 3_000_001, 256, 64)          # > 2^31 bytes of x? (768 MB) and odd M
check(1, 9_000_000, 64, 64)           # 2.3 GB of x: 64-bit offsets
check(1, 1000, 100_003, 32)           # large odd K, split-K
check(64, 100, 1000, 16)              # many heads, tiny M
check(3, 50_000, 5000, 200, metric=1) # padded D, dot
check(1, 300, 40, 512)                # Dp = 512 small
check(2, 10_000, 300, 384)            # D padded to 512
