"""Container-only: the reference and the drop-in as two synchronised replicas (gloo, 2 ranks, different data per rank):
EMA statistics all-reduce, k-means seeding with distributed sampling, dead-code re-seeding (distributed and averaged).
Per rank the drop-in must match the reference; across ranks the codebooks must stay identical.

    python tests/golden/sweep_distributed.py
"""
from __future__ import annotations

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.argv = [sys.argv[0]]
    sys.path.insert(0, HERE)
    import sweep_against_reference as sw

    ref, ref_cb, mine = sw.ref, sw.ref_cb, sw.mine
    mine_params = sys.modules["vq_dropin.params"]
    cases = {
        "ema": dict(cb=dict(dim=16, codebook_size=24, threshold_ema_dead_code=0, kmeans=None)),
        "kmeans": dict(cb=dict(dim=16, codebook_size=12, threshold_ema_dead_code=0, initialization_by_kmeans=True, kmeans=(4, True))),
        "expire_all_ranks": dict(cb=dict(dim=16, codebook_size=200, threshold_ema_dead_code=2, kmeans=(4, True))),
        "expire_averaged": dict(cb=dict(dim=16, codebook_size=200, threshold_ema_dead_code=2, distributed_replace_codes=False,
                                        kmeans=(4, True))),
    }
    report = []
    for name, spec in cases.items():
        cb_kw = dict(spec["cb"])
        km = cb_kw.pop("kmeans")
        kr, kmi = dict(cb_kw), dict(cb_kw)
        if km is not None:
            kr["kmeans_params"] = ref_cb.KmeansParameters(iter=km[0], sync=km[1])
            kmi["kmeans_params"] = mine_params.KmeansParameters(iter=km[0], sync=km[1])
        else:  # the fork's constructor needs kmeans_params under DDP
            kr["kmeans_params"] = ref_cb.KmeansParameters()
            kmi["kmeans_params"] = mine_params.KmeansParameters()
        torch.manual_seed(5)
        r = ref.VectorQuantize(dim=16, codebook_params=ref_cb.CodebookParams(**kr), sync_codebook=True)
        m = mine.VectorQuantize(dim=16, codebook_params=sw.MineParams(**kmi), sync_codebook=True)
        m.load_state_dict(r.state_dict())
        r.train()
        m.train()
        ok = True
        for step in range(2):
            x = torch.randn(3 + rank, 20, 16, generator=torch.Generator().manual_seed(100 * step + rank))
            with torch.no_grad():
                torch.manual_seed(1000 + step + 17 * rank)
                a = r(x)
                torch.manual_seed(1000 + step + 17 * rank)
                b = m(x)
            try:
                sw.compare(f"{name} step {step} out", b, a)
                for k, v in r.state_dict().items():
                    sw.compare(f"{name} step {step} state[{k}]", m.state_dict()[k], v, tol=1e-4)
            except AssertionError as e:
                ok = False
                report.append(f"rank {rank} {name}: DEVIATION {e}")
                break
        # replicas must hold the same codebook
        mine_cb = m._codebook.embeddings.detach().clone()
        gathered = [torch.empty_like(mine_cb) for _ in range(world)]
        dist.all_gather(gathered, mine_cb)
        same = all(torch.allclose(g, gathered[0], atol=1e-6) for g in gathered)
        report.append(f"rank {rank} {name}: {'AGREE' if ok else 'DEVIATION'}; replicas identical: {same}")
    with open(os.path.join(out, f"r{rank}.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        for r in range(2):
            print(open(os.path.join(out, f"r{r}.txt")).read(), end="")


if __name__ == "__main__":
    main()
