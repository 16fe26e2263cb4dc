"""Per-rank cost of the K-sharded search at N ranks, emulated on one GPU (no collective): shard = K/N codes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch
from vector_quantization.sharded import ShardedCodebookSearch
K, D, M = 65536, 512, 8192
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(99)
full = torch.randn((K, D), generator=g).to(dev)
x = torch.randn((M, D), generator=g).to(dev)
for n in (1, 2, 4, 8):
    s = ShardedCodebookSearch(full[: K // n].contiguous(), full_codebook=full)
    for _ in range(3):
        s(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        s(x)
    e1.record(); torch.cuda.synchronize()
    print(f"N={n}: shard K={K//n}: wall {ms:.3f} ms/step, gpu {e0.elapsed_time(e1)/20:.3f} ms/step -> x{'%.2f' % (1.0)}")
