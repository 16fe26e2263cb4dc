"""Three inference forwards of VectorQuantize (cfg2) and of ResidualVQ (cfg4) after a warm-up, for
`rocprofv3 --kernel-trace --output-format csv -- python3 tools/one_forward.py`: after the first call of each module (which
packs the codebooks) every forward is ONE kernel dispatch (profiles/r02_one_forward_kernel_trace.csv)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")]
import torch
import vector_quantization as vq
from vector_quantization.codebooks import CodebookParams

dev = "cuda:0"
torch.manual_seed(0)
m1 = vq.VectorQuantize(dim=256, codebook_params=CodebookParams(dim=256, codebook_size=1024)).to(dev).eval()
m2 = vq.ResidualVQ(dim=256, num_quantizers=8, codebook_params=CodebookParams(dim=256, codebook_size=1024)).to(dev).eval()
x1 = torch.randn(256, 1024, 256, device=dev)
x2 = torch.randn(64, 1024, 256, device=dev)
torch.cuda.synchronize()
with torch.no_grad():
    for mod, x in ((m1, x1), (m2, x2)):
        mod(x)  # warm-up: packs the codebooks (vq_pack_kernel) and fills the module's caches
        torch.cuda.synchronize()
        for _ in range(3):
            mod(x)
            torch.cuda.synchronize()
