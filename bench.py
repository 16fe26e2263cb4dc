#!/usr/bin/env python3
"""bench.py -- vectors quantized / second on the nearest-codebook hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|k8192|cfg3a|cfg4|cfg5]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
(one rank per GPU, RCCL).  A "step" is one full eval-mode module forward (codebook pack + fused
search/gather launch) over one batch of synthetic input already resident in HBM.  Rank 0 prints ONE JSON line.

Default workload = BASELINE configs[1]: VectorQuantize(dim=256, codebook_size=1024) on a [256, 1024, 256]
batch per GPU (M = 262 144 rows per GPU; rows shard over ranks with no data-path collective -> weak scaling).

Extra objects in the JSON line:
  roofline      dominant kernel (vq_search_mfma) priced against the fp32 MFMA peak: algorithmic FLOPs
                (2*K*D per searched row) / mean launch duration measured with HIP events on the launch stream.
  cpu_baseline  the reference's ATen op sequence (oracle/ref_path.py, "port") timed on the host cores on a bounded
                sample, rank 0 / N=1 only.
  sharded_k65536  (informational) codebook K=65536, D=512 sharded over the N ranks with the packed-key
                MIN all-reduce (BASELINE configs[4]); at N=1 this is the full-codebook 1-GPU figure.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vector-quantization-by-ml_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 256 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (kind, dim, K, batch shape, heads, Q)
    "cfg2": dict(kind="vq", dim=256, K=1024, x_shape=(256, 1024, 256), desc="VectorQuantize dim=256 codebook_size=1024, batch [256,1024,256] per GPU, eval"),
    "k8192": dict(kind="vq", dim=256, K=8192, x_shape=(256, 1024, 256), desc="VectorQuantize dim=256 codebook_size=8192, batch [256,1024,256] per GPU, eval (north-star roofline shape)"),
    "cfg3a": dict(kind="vq", dim=512, K=8192, x_shape=(64, 1024, 512), heads=8, codebook_dim=64, desc="VectorQuantize dim=512 heads=8 codebook_dim=64 per-head codebooks K=8192, batch [64,1024,512] per GPU, eval"),
    "cfg4": dict(kind="rvq", dim=256, K=1024, Q=8, x_shape=(64, 1024, 256), desc="ResidualVQ num_quantizers=8 codebook_size=1024 dim=256, batch [64,1024,256] per GPU, eval"),
    "cfg1": dict(kind="vq", dim=64, K=256, x_shape=(32, 256, 64), desc="VectorQuantize dim=64 codebook_size=256, batch [32,256,64], eval"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_module(w, device, seed=4321):
    import vector_quantization as vq
    from vector_quantization.codebooks import CodebookParams

    g = torch.Generator().manual_seed(seed)
    if w["kind"] == "vq":
        heads = w.get("heads", 1)
        cd = w.get("codebook_dim", None)
        d = cd if cd is not None else w["dim"]
        mod = vq.VectorQuantize(dim=w["dim"], codebook_params=CodebookParams(dim=d, codebook_size=w["K"]), heads=heads,
                                codebook_dim=cd, separate_codebook_per_head=heads > 1)
        with torch.no_grad():
            mod._codebook.embeddings.copy_(torch.randn(mod._codebook.embeddings.shape, generator=g))
    else:
        mod = vq.ResidualVQ(dim=w["dim"], num_quantizers=w["Q"], codebook_params=CodebookParams(dim=w["dim"], codebook_size=w["K"]))
        with torch.no_grad():
            for i, layer in enumerate(mod.layers):
                layer._codebook.embeddings.copy_(torch.randn((1, w["K"], w["dim"]), generator=g) * 2.0 ** (-i / 2.0))
    return mod.to(device).eval()


def rows_per_step(w):
    n = 1
    for s in w["x_shape"][:-1]:
        n *= s
    return n * w.get("heads", 1) if w["kind"] == "vq" else n


def flops_per_step(w):
    heads = w.get("heads", 1)
    d = w.get("codebook_dim", None) or w["dim"]
    tokens = 1
    for s in w["x_shape"][:-1]:
        tokens *= s
    return 2.0 * tokens * heads * w["K"] * d * w.get("Q", 1)


def kernel_roofline(w, device, mod, x):
    """Mean duration of the dominant kernel (vq_search_mfma) alone, HIP events on the launch stream."""
    from vector_quantization import native

    heads = w.get("heads", 1)
    if w["kind"] == "vq":
        cb = mod._codebook.embeddings.detach()[:, None].contiguous()  # [H, 1, K, D]
        d = cb.shape[-1]
        rows = x.numel() // (heads * d)
        flat = x.reshape(rows, heads, d).permute(1, 0, 2)
    else:
        cb = torch.stack([layer._codebook.embeddings.detach()[0] for layer in mod.layers])[None].contiguous()
        flat = x.reshape(1, -1, x.shape[-1])
    packed = native.pack_codebooks(cb, native.EUCLID)
    out = torch.empty(flat.shape, dtype=torch.float32, device=device)
    idx = torch.empty((flat.shape[0], flat.shape[1], cb.shape[1]), dtype=torch.int64, device=device)
    for _ in range(3):
        native.quantize(flat, cb, packed=packed, want_best=False, out=out, idx=idx)
    torch.cuda.synchronize(device)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        native.quantize(flat, cb, packed=packed, want_best=False, out=out, idx=idx)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / n
    fl = flops_per_step(w)
    achieved = fl / (ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(w["name"], {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, kernel="vq_search_mfma",
                kernel_ms=round(ms, 4), algorithmic_flops_per_launch=fl,
                algorithmic_hbm_bytes_per_launch=(8 * (w.get("codebook_dim") or w["dim"]) + 8 * w.get("Q", 1)) * rows_per_step(w))


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(w, budget_s):
    """The reference's ATen op sequence on the host cores, bounded sample of the same workload."""
    from oracle import ref_path

    cores = usable_cores()
    torch.set_num_threads(cores)

    dim, K = w["dim"], w["K"]
    g = torch.Generator().manual_seed(1234)
    if w["kind"] == "rvq":
        sample_shape = (4, 1024, dim)
        cbs = torch.stack([torch.randn((K, dim), generator=g) * 2.0 ** (-i / 2.0) for i in range(w["Q"])])
        x = torch.randn(sample_shape, generator=g)
        fn = lambda: ref_path.residual_vq_forward(x, cbs)  # noqa: E731
    else:
        heads = w.get("heads", 1)
        d = w.get("codebook_dim", None) or dim
        b = max(1, min(w["x_shape"][0], 65536 // w["x_shape"][1]))
        sample_shape = (b, w["x_shape"][1], dim)
        cb = torch.randn((heads, K, d), generator=g)
        x = torch.randn(sample_shape, generator=g)

        def fn():
            flat = x.reshape(-1, heads, d).permute(1, 0, 2)
            return ref_path.codebook_forward(flat, cb)
    rows = sample_shape[0] * sample_shape[1] * (w.get("heads", 1) if w["kind"] == "vq" else 1)
    fn()  # warm-up
    t0 = time.perf_counter()
    it = 0
    while True:
        fn()
        it += 1
        el = time.perf_counter() - t0
        if el >= budget_s or it >= 200:
            break
    return dict(value=round(rows * it / el, 1), unit="vectors/s", cores=cores, kind="port",
                sample=f"oracle/ref_path.py (reference ATen op sequence: -cdist, argmax, one_hot, gather) on x{list(sample_shape)}, "
                       f"{it} iterations in {el:.1f} s, torch {torch.__version__} CPU, {cores} threads "
                       f"(machine reports {os.cpu_count()} cpus)")


def parity_gate(w, mod, x):
    """Part of the cpu_baseline leg (rank 0, N = 1): the CPU oracle is the CHECKER for a sample of the GPU result."""
    from oracle import vq_oracle

    with torch.no_grad():
        sample = x[:2]
        out_s = mod(sample)
    if w["kind"] == "vq" and w.get("heads", 1) == 1:
        ref = vq_oracle.vq_forward(sample.reshape(1, -1, w["dim"]).cpu().numpy(), mod._codebook.embeddings.cpu().numpy())
        ok = bool((out_s[1].reshape(-1).cpu().numpy() == ref["idx"][0]).all())
        n_s = ref["idx"].size
    elif w["kind"] == "rvq":
        cbs = torch.stack([layer._codebook.embeddings[0] for layer in mod.layers]).cpu().numpy()
        ref = vq_oracle.rvq_forward(sample.reshape(-1, w["dim"]).cpu().numpy(), cbs)
        ok = bool((out_s[1].reshape(-1, w["Q"]).cpu().numpy() == ref["idx"]).all())
        n_s = ref["idx"].size
    else:
        h, d = w["heads"], w["codebook_dim"]
        flat = sample.reshape(-1, h, d).permute(1, 0, 2).contiguous().cpu().numpy()
        ref = vq_oracle.vq_forward(flat, mod._codebook.embeddings.cpu().numpy())
        ok = bool((out_s[1].reshape(-1, h).cpu().numpy() == ref["idx"].T).all())
        n_s = ref["idx"].size
    if not ok:
        raise SystemExit("PARITY FAILURE: GPU indices differ from the CPU oracle; refusing to report a number")
    return f"indices bit-exact vs CPU oracle on a {n_s}-index sample"


def sharded_k65536(device, rank, world, steps=5):
    """BASELINE configs[4]: K=65536, D=512 sharded over the ranks, packed-key MIN all-reduce over RCCL."""
    from vector_quantization.sharded import ShardedCodebookSearch

    K, D, M = 65536, 512, 8192
    g = torch.Generator().manual_seed(99)
    kl = K // world
    full = torch.randn((K, D), generator=g)
    x = torch.randn((M, D), generator=torch.Generator().manual_seed(1234)).to(device)
    shard = full[rank * kl:(rank + 1) * kl].to(device)
    table = full.to(device)  # replicated gather table (128 MiB)
    s = ShardedCodebookSearch(shard, full_codebook=table)
    for _ in range(2):
        s(x)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out, idx, best, _ = s(x)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    return dict(value=round(M * steps / el, 1), unit="vectors/s", n_gpus=world, ms_per_step=round(el / steps * 1e3, 3),
                config="K=65536 D=512 M=8192 tokens replicated, codebook sharded K/N per GPU, 8-byte key MIN all-reduce, replicated gather table",
                idx_checksum=int(idx.sum().item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)  # the chip needs ~30 ms of load to settle its clock
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed load before the W warm-up steps so the chip's clock has settled (DVFS ramp ~30 ms)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal knobs (1-GPU box): all ranks on cuda:0 over gloo, to exercise the N > 1 code path
    one_device = os.environ.get("VQ_BENCH_ONE_DEVICE", "0") == "1"
    backend = os.environ.get("VQ_BENCH_BACKEND", "nccl")
    device = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from vector_quantization import native

    native.load()
    w = dict(WORKLOADS[args.workload], name=args.workload)
    mod = build_module(w, device)
    x = torch.randn(w["x_shape"], generator=torch.Generator().manual_seed(1234 + rank)).to(device)

    parity = None

    # ---- timed region ------------------------------------------------------------------------------------------
    with torch.no_grad():
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:  # not part of W or K: lets the clock ramp up
            mod(x)
            torch.cuda.synchronize(device)
        for _ in range(args.warmup):
            mod(x)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        native.begin_kernel_timing()  # HIP events around the search launch, on its own stream, inside the timed region
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mod(x)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
        kernel_events = native.end_kernel_timing()
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_rows = rows_per_step(w) * args.steps * world
    value = total_rows / elapsed

    roof = None
    with torch.no_grad():
        roof = kernel_roofline(w, device, mod, x)
    if kernel_events:
        # the figure of record: mean duration of the search launch DURING the timed steps
        live_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / len(kernel_events)
        roof["kernel_ms_isolated"] = roof["kernel_ms"]
        roof["kernel_ms"] = round(live_ms, 4)
        roof["achieved"] = round(roof["algorithmic_flops_per_launch"] / (live_ms * 1e-3) / 1e12, 2)
        roof["frac"] = round(roof["achieved"] / PEAK_F32_MFMA_TFLOPS, 4)
        roof["launches_timed"] = len(kernel_events)
    sharded = None
    if not args.no_sharded:
        try:
            with torch.no_grad():
                sharded = sharded_k65536(device, rank, world)
        except Exception as e:  # informational leg: never fail the bench line
            sharded = dict(error=str(e)[:200])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        parity = parity_gate(w, mod, x)
        cpu = cpu_baseline(w, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "vectors quantized/sec on [B*N,D]x[K,D] argmin; indices bit-exact vs CPU",
            "value": round(value, 1),
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w["desc"], "name": args.workload, "rows_per_gpu_per_step": rows_per_step(w),
                       "sharding": "rows (tokens) sharded over ranks, codebook replicated, no data-path collective"},
            "parity": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
            "sharded_k65536": sharded,
            "device": native.device_info(),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
