"""Eager cfg1 forwards for `rocprofv3 --kernel-trace --stats -- python3 tools/cfg1_prof.py` (per-kernel durations at the
reference's CPU-sized configuration)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch
import vector_quantization as vq
from vector_quantization.codebooks import CodebookParams
dev = "cuda:0"
mod = vq.VectorQuantize(dim=64, codebook_params=CodebookParams(dim=64, codebook_size=256)).to(dev).eval()
x = torch.randn(32, 256, 64, device=dev)
with torch.no_grad():
    for _ in range(50):
        mod(x)
torch.cuda.synchronize()
