"""DDP codebook synchronisation (reference: codebooks.py:410,415 all_reduce of the EMA statistics) on 2 gloo ranks:
each rank quantises HALF of a batch; after one training forward both replicas must hold the codebook a single
process obtains from the WHOLE batch (CPU, checker backend for the search)."""
from __future__ import annotations

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(sync):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams
    from gen import make_codebook

    mod = vq.VectorQuantize(dim=16, codebook_params=CodebookParams(dim=16, codebook_size=32, threshold_ema_dead_code=0),
                            sync_codebook=sync)
    cb = make_codebook(1, 32, 16, "S")
    with torch.no_grad():
        mod._codebook.embeddings.copy_(cb)
        mod._codebook.embed_avg.copy_(cb)
    return mod.train()


def _worker(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "vector-quantization-by-ml_amd"), os.path.join(root, "tests"),
              os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gen import make_x
    from helpers import OracleBackend
    from vector_quantization import search

    search.set_backend(OracleBackend)
    mod = _make(sync=None)  # default: sync when a process group with > 1 ranks exists
    assert mod._codebook.use_ddp
    x = make_x((4, 50, 16), "S")
    half = x[rank * 2:(rank + 1) * 2]
    with torch.no_grad():
        mod(half)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), emb=mod._codebook.embeddings.numpy(),
             cs=mod._codebook.cluster_size.numpy(), avg=mod._codebook.embed_avg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_match_single_process(tmp_path, oracle):
    from gen import make_x
    from helpers import OracleBackend
    from vector_quantization import search

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    search.set_backend(OracleBackend)
    try:
        mod = _make(sync=False)
        with torch.no_grad():
            mod(make_x((4, 50, 16), "S"))
    finally:
        search.set_backend(None)
    for r in range(2):
        z = np.load(tmp_path / f"r{r}.npz")
        np.testing.assert_allclose(z["cs"], mod._codebook.cluster_size.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(z["avg"], mod._codebook.embed_avg.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["emb"], mod._codebook.embeddings.numpy(), rtol=1e-5, atol=1e-6)


def _worker_seeding(rank, world, port, out_dir, distributed_replace):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "vector-quantization-by-ml_amd"), os.path.join(root, "tests"),
              os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vector_quantization as vq
    from helpers import OracleBackend
    from vector_quantization import search
    from vector_quantization.codebooks import CodebookParams, KmeansParameters

    search.set_backend(OracleBackend)
    torch.manual_seed(3)  # same module initialisation on every replica
    params = CodebookParams(dim=16, codebook_size=150, initialization_by_kmeans=True,
                            kmeans_params=KmeansParameters(iter=3, sync=True), threshold_ema_dead_code=2,
                            distributed_replace_codes=distributed_replace)
    mod = vq.VectorQuantize(dim=16, codebook_params=params).train()
    assert mod._codebook.use_ddp
    for step in range(3):
        torch.manual_seed(50 + step + 7 * rank)  # DIFFERENT data and generator state per replica
        with torch.no_grad():
            mod(torch.randn(3 + rank, 30, 16))
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), emb=mod._codebook.embeddings.numpy(),
             cs=mod._codebook.cluster_size.numpy(), avg=mod._codebook.embed_avg.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_stay_identical_through_kmeans_seeding_and_dead_code_reseeding(tmp_path, oracle):
    """k-means seeding from the first batch and dead-code re-seeding draw their vectors from ALL replicas' rows
    (utils/distributed.py:56-78) -- or average the replicas' draws (codebooks.py:236-237) -- so that replicas fed with
    different data keep bit-identical codebooks; with 150 codes and ~100 rows per step most codes expire every step."""
    for distributed_replace in (True, False):
        port = _free_port()
        mp.spawn(_worker_seeding, args=(2, port, str(tmp_path), distributed_replace), nprocs=2, join=True)
        a, b = np.load(tmp_path / "s0.npz"), np.load(tmp_path / "s1.npz")
        for key in ("emb", "cs", "avg"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{key} (distributed_replace_codes={distributed_replace})")
        assert np.isfinite(a["emb"]).all() and float(np.abs(a["emb"]).max()) > 0


def _worker_gpu(rank, world, port, out_dir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "vector-quantization-by-ml_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two processes share the one GPU of the test box
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams, KmeansParameters

    dev = "cuda:0"
    torch.manual_seed(3)
    params = CodebookParams(dim=32, codebook_size=200, initialization_by_kmeans=True,
                            kmeans_params=KmeansParameters(iter=3, sync=True), threshold_ema_dead_code=2)
    mod = vq.VectorQuantize(dim=32, codebook_params=params).to(dev).train()
    assert mod._codebook.use_ddp
    for step in range(3):
        torch.manual_seed(50 + step + 7 * rank)
        with torch.no_grad():
            mod(torch.randn(3 + rank, 40, 32, device=dev))
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), emb=mod._codebook.embeddings.cpu().numpy(),
             cs=mod._codebook.cluster_size.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_native_replicas_stay_identical_on_the_gpu(tmp_path):
    """The same replica-consistency property with the NATIVE kernels (EMA statistics, search) on the device: two
    processes on the one GPU, collectives over gloo."""
    port = _free_port()
    mp.spawn(_worker_gpu, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "g0.npz"), np.load(tmp_path / "g1.npz")
    np.testing.assert_array_equal(a["emb"], b["emb"])
    np.testing.assert_array_equal(a["cs"], b["cs"])
