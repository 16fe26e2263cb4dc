import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
from oracle import vq_oracle as o
from gen import make_x, make_codebook
from vector_quantization import native

for (H, M, K, D, metric) in [(1, 8192, 256, 64, 0), (1, 4096, 1024, 256, 0), (1, 8192, 256, 64, 1), (1, 111, 300, 100, 0)]:
    x = make_x((H, M, D)); cb = make_codebook(H, K, D)
    ref = o.vq_forward(x.numpy(), cb.numpy(), metric)
    for flags in (0, native.F_FORCE_SIMPLE):
        r = native.quantize(x.cuda(), cb[:, None].contiguous().cuda(), metric=metric, flags=flags, want_sq_err=True)
        torch.cuda.synchronize()
        gb = r["best"][..., 0].cpu().numpy(); rb = ref["best"]
        gi = r["idx"][..., 0].cpu().numpy()
        bad = gb.view(np.uint32) != rb.view(np.uint32)
        ulp = (gb.view(np.int32).astype(np.int64) - rb.view(np.int32).astype(np.int64))
        print(f"H{H} M{M} K{K} D{D} metric{metric} flags{flags}: idx_mismatch={(gi != ref['idx']).sum()} best_mismatch={bad.sum()} "
              f"ulp min/max={ulp.min()}/{ulp.max()}  sq_err {float(r['sq_err'][0]):.6f} vs {ref['sq_err']:.6f}")
        if bad.any():
            j = np.argwhere(bad)[:5]
            for (h, m) in j:
                print("   row", m, "gpu", gb[h, m], hex(gb.view(np.uint32)[h, m]), "ref", rb[h, m], hex(rb.view(np.uint32)[h, m]), "sq gpu", float(gb[h,m])**2)
