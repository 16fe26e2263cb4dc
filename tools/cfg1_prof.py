import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/vector-quantization-by-ml_amd"]
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
