#!/usr/bin/env python3
"""bench.py -- vectors quantized / second on the nearest-codebook hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|k8192|cfg3a|cfg3b|cfg4|cfg1|wide1024] [--legs a,b,...]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
(one rank per GPU, RCCL).  A "step" is one full eval-mode module forward (fused search/gather launch; the packed
codebook image is cached by the module) over one batch of synthetic input already resident in HBM.  Rank 0 prints
ONE JSON line.

Headline workload = BASELINE configs[1]: VectorQuantize(dim=256, codebook_size=1024) on a [256, 1024, 256] batch per
GPU (M = 262 144 rows per GPU; rows shard over ranks with no data-path collective -> weak scaling).

Objects in the JSON line beside the contract's keys:
  roofline      dominant kernel (vq_search_mfma) priced against the fp32 MFMA peak: algorithmic FLOPs (2*K*D per
                searched row and stage) / mean launch duration measured with HIP events on the launch stream INSIDE
                the timed region.  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of the same command
                (profiles/traffic.json; gfx950 correction of MI355X_MICROARCH.md applied), null if not collected.
  cpu_baseline  the reference's ATen op sequence (oracle/ref_path.py, "port") timed on the host cores on a bounded
                sample, rank 0 / N = 1 only; also the 1-thread figure and the CPU model.
  legs          (N = 1) the other single-GPU BASELINE workloads, each run like the headline (same W / K): k8192
                (north-star roofline shape), cfg3a / cfg3b (multi-head, per-head dim 64 / 512), cfg4 (ResidualVQ) --
                value, ms_per_step, roofline and a CPU-oracle parity gate per leg; plus wide1024 (rows of 1024 dims:
                not a BASELINE config, the sliced sweep for rows wider than one launch holds).
  sharded_k65536  BASELINE configs[4]: K = 65536, D = 512 sharded over the N ranks (packed-key planes, MIN all-reduce or
                one-hop all-gather + MIN in the finalize), M = 8192 and M = 65536, each with per-phase times (search /
                exchange / finalize / host enqueue); at N = 1 the full-codebook 1-GPU figure of the same kernels.
"""
from __future__ import annotations

import argparse
import gc
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
    "cfg2": dict(kind="vq", dim=256, K=1024, x_shape=(256, 1024, 256),
                 desc="VectorQuantize dim=256 codebook_size=1024, batch [256,1024,256] per GPU, eval"),
    "k8192": dict(kind="vq", dim=256, K=8192, x_shape=(256, 1024, 256),
                  desc="VectorQuantize dim=256 codebook_size=8192, batch [256,1024,256] per GPU, eval (north-star roofline shape)"),
    "cfg3a": dict(kind="vq", dim=512, K=8192, x_shape=(64, 1024, 512), heads=8, codebook_dim=64,
                  desc="VectorQuantize dim=512 heads=8 codebook_dim=64 per-head codebooks K=8192, batch [64,1024,512] per GPU, eval"),
    "cfg3b": dict(kind="vq", dim=512, K=8192, x_shape=(64, 1024, 512), heads=8, codebook_dim=512,
                  desc="VectorQuantize dim=512 heads=8 codebook_dim=512 (Linear 512->4096->512 projections) per-head codebooks "
                       "K=8192, batch [64,1024,512] per GPU, eval; value = whole module forward, roofline = search kernel only"),
    "cfg4": dict(kind="rvq", dim=256, K=1024, Q=8, x_shape=(64, 1024, 256),
                 desc="ResidualVQ num_quantizers=8 codebook_size=1024 dim=256, batch [64,1024,256] per GPU, eval"),
    # cfg4's stack at row counts that are NOT a multiple of a round of workgroups (not BASELINE configs; round 3's launch plans):
    "cfg4_m70000": dict(kind="rvq", dim=256, K=1024, Q=8, x_shape=(70, 1000, 256),
                        kernel_label="vq_search_mfma<.., MULTI> on the rows that fill whole rounds + the remaining 4464 rows stage by stage "
                                     "(K-split search + finalize per stage); roofline = the whole launch sequence of one call",
                        desc="ResidualVQ num_quantizers=8 codebook_size=1024 dim=256, batch [70,1000,256] per GPU (70 000 rows: 1.07 rounds "
                             "of 256-row workgroups), eval"),
    "cfg4_m8192": dict(kind="rvq", dim=256, K=1024, Q=8, x_shape=(8, 1024, 256),
                       kernel_label="every stage a K-split search over all CUs + finalize (few rows: no fused part); roofline = the whole "
                                    "launch sequence of one call",
                       desc="ResidualVQ num_quantizers=8 codebook_size=1024 dim=256, batch [8,1024,256] per GPU (8192 rows: 32 row blocks "
                            "on 256 CUs), eval"),
    "cfg1": dict(kind="vq", dim=64, K=256, x_shape=(32, 256, 64), desc="VectorQuantize dim=64 codebook_size=256, batch [32,256,64], eval"),
    # not a BASELINE config: rows wider than the 512 dims one launch holds (the reference has no width limit); the search is a
    # sequence of launches here (256-dim slices, chains carried through the workspace, then the finalize), timed as one
    "wide1024": dict(kind="vq", dim=1024, K=1024, x_shape=(64, 1024, 1024),
                     desc="VectorQuantize dim=1024 codebook_size=1024, batch [64,1024,1024] per GPU, eval (rows wider than 512 dims: "
                          "sliced sweep; roofline = the whole launch sequence of one search)"),
}
DEFAULT_LEGS = "k8192,cfg3a,cfg3b,cfg4,wide1024,cfg4_m70000,cfg4_m8192"


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
        torch.manual_seed(seed)  # projection weights of cfg3b
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


def tokens_per_step(w):
    n = 1
    for s in w["x_shape"][:-1]:
        n *= s
    return n


def rows_per_step(w):
    return tokens_per_step(w) * (w.get("heads", 1) if w["kind"] == "vq" else 1)


def head_dim(w):
    return w.get("codebook_dim", None) or w["dim"]


def flops_per_step(w):
    return 2.0 * tokens_per_step(w) * w.get("heads", 1) * w["K"] * head_dim(w) * w.get("Q", 1)


def algorithmic_bytes_per_step(w):
    """per searched row read x, write q, write one int64 per stage: 8 D + 8 Q (+ the codebook once)."""
    return (8 * head_dim(w) + 8 * w.get("Q", 1)) * rows_per_step(w)


def traffic_record(name):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def isolated_kernel_ms(w, device, mod, x, n=10):
    """Back-to-back launches of the search kernel alone through the C ABI (informational cross-check of the live figure)."""
    from vector_quantization import native

    heads = w.get("heads", 1)
    if w["kind"] == "vq":
        cb = mod._codebook.embeddings.detach()[:, None].contiguous()  # [H, 1, K, D]
        d = cb.shape[-1]
        xin = mod.project_in(x)
        rows = xin.numel() // (heads * d)
        flat = xin.reshape(rows, heads, d).permute(1, 0, 2)
    else:
        cb = torch.stack([layer._codebook.embeddings.detach()[0] for layer in mod.layers])[None].contiguous()
        flat = x.reshape(1, -1, x.shape[-1])
    packed = native.pack_codebooks(cb, native.EUCLID)
    out = torch.empty(flat.shape, dtype=torch.float32, device=device)
    idx = torch.empty((flat.shape[0], flat.shape[1], cb.shape[1]), dtype=torch.int64, device=device)
    for _ in range(3):
        native.quantize(flat, cb, packed=packed, want_best=False, out=out, idx=idx)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        native.quantize(flat, cb, packed=packed, want_best=False, out=out, idx=idx)
    e1.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) / n


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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(w, budget_s):
    """The reference's ATen op sequence on the host cores, bounded sample of the same workload; all usable threads
    (the reported value) and one thread."""
    from oracle import ref_path

    cores = usable_cores()
    dim, K = w["dim"], w["K"]
    g = torch.Generator().manual_seed(1234)
    if w["kind"] == "rvq":
        sample_shape = (4, 1024, dim)
        cbs = torch.stack([torch.randn((K, dim), generator=g) * 2.0 ** (-i / 2.0) for i in range(w["Q"])])
        x = torch.randn(sample_shape, generator=g)
        fn = lambda: ref_path.residual_vq_forward(x, cbs)  # noqa: E731
    else:
        heads = w.get("heads", 1)
        d = head_dim(w)
        b = max(1, min(w["x_shape"][0], 65536 // w["x_shape"][1]))
        sample_shape = (b, w["x_shape"][1], heads * d)
        cb = torch.randn((heads, K, d), generator=g)
        x = torch.randn(sample_shape, generator=g)

        def fn():
            flat = x.reshape(-1, heads, d).permute(1, 0, 2)
            return ref_path.codebook_forward(flat, cb)
    rows = sample_shape[0] * sample_shape[1] * (w.get("heads", 1) if w["kind"] == "vq" else 1)

    def timed(threads, budget):
        torch.set_num_threads(threads)
        fn()  # warm-up
        t0 = time.perf_counter()
        it = 0
        while True:
            fn()
            it += 1
            el = time.perf_counter() - t0
            if el >= budget or it >= 200:
                break
        return rows * it / el, it, el

    v_all, it, el = timed(cores, budget_s)
    v_one, it1, el1 = timed(1, max(2.0, budget_s / 2.0))
    torch.set_num_threads(cores)
    return dict(value=round(v_all, 1), unit="vectors/s", cores=cores, kind="port", one_thread_value=round(v_one, 1),
                cpu_model=cpu_model(),
                sample=f"oracle/ref_path.py (reference ATen op sequence: -cdist, argmax, one_hot, gather) on x{list(sample_shape)}: "
                       f"{it} iterations in {el:.1f} s with {cores} threads, {it1} in {el1:.1f} s with 1 thread; "
                       f"torch {torch.__version__} CPU on {cpu_model()} (machine reports {os.cpu_count()} cpus)")


def parity_gate(w, mod, x):
    """rank 0, N = 1: the CPU oracle is the CHECKER for a sample of the GPU result (never the thing measured)."""
    from oracle import vq_oracle

    kind, heads = w["kind"], w.get("heads", 1)
    with torch.no_grad():
        if kind == "vq" and heads == 1:
            sample = x[:2]
            got = mod(sample)[1].reshape(-1).cpu().numpy()
            ref = vq_oracle.vq_forward(sample.reshape(1, -1, w["dim"]).cpu().numpy(), mod._codebook.embeddings.cpu().numpy())
            ok, n_s = bool((got == ref["idx"][0]).all()), ref["idx"].size
        elif kind == "rvq":
            sample = x[:2]
            got = mod(sample)[1].reshape(-1, w["Q"]).cpu().numpy()
            cbs = torch.stack([layer._codebook.embeddings[0] for layer in mod.layers]).cpu().numpy()
            ref = vq_oracle.rvq_forward(sample.reshape(-1, w["dim"]).cpu().numpy(), cbs)
            ok, n_s = bool((got == ref["idx"]).all()), ref["idx"].size
        else:
            h, d = heads, head_dim(w)
            sample = x[:2] if d <= 64 else x[:1, :256]
            got = mod(sample)[1].reshape(-1, h).cpu().numpy()
            # the rows the search sees: after the (GPU) input projection, split into heads
            flat = mod.project_in(sample).reshape(-1, h, d).permute(1, 0, 2).contiguous().cpu().numpy()
            ref = vq_oracle.vq_forward(flat, mod._codebook.embeddings.cpu().numpy())
            ok, n_s = bool((got == ref["idx"].T).all()), ref["idx"].size
    if not ok:
        raise SystemExit(f"PARITY FAILURE ({w['name']}): GPU indices differ from the CPU oracle; refusing to report a number")
    return f"indices bit-exact vs CPU oracle on a {n_s}-index sample"


def run_workload(name, args, device, rank, world, want_parity):
    """W warm-up + K timed steps of one workload; -> (value rows/s over all ranks, ms_per_step, roofline, parity, w)."""
    from vector_quantization import native

    w = dict(WORKLOADS[name], name=name)
    mod = build_module(w, device)
    x = torch.randn(w["x_shape"], generator=torch.Generator().manual_seed(1234 + rank)).to(device)
    with torch.no_grad():
        # not part of W or K: ~60 ms of the workload itself let the chip's clock ramp up (DVFS).  (Round 3 tried a library
        # GEMM here so that a profiler's per-kernel average would hold steady-state launches only: the 1-ms search launches
        # that follow a GEMM burst run 7 % SLOWER for tens of milliseconds -- profiles/r03 notes -- so the settle phase stays
        # the search kernel, and `rocprofv3 --stats` averages of this command include its ~55 ramp-up launches.)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
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
    value = rows_per_step(w) * args.steps * world / elapsed

    fl = flops_per_step(w)
    live_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(1, len(kernel_events))
    with torch.no_grad():
        iso_ms = isolated_kernel_ms(w, device, mod, x)
    achieved = fl / (live_ms * 1e-3) / 1e12
    ab = algorithmic_bytes_per_step(w)
    roof = dict(bound="mfma", achieved=round(achieved, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic_record(name),
                traffic_source="profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this workload's bench command "
                               "(tools/profile_round.sh), per search; a recorded measurement, not taken during this run",
                kernel=(w["kernel_label"] if "kernel_label" in w else
                        "vq_search_pair512<.., WIDE> x 512-dim slices (the last one finishes the call)" if head_dim(w) > 512 else
                        "vq_search_pair512" if (head_dim(w) > 256 and w.get("Q", 1) == 1) else
                        "vq_search_persist" if (128 < head_dim(w) <= 256 and w.get("Q", 1) == 1 and 1024 <= w["K"] <= 3072) else
                        "vq_search_mfma"),
                kernel_ms=round(live_ms, 4), kernel_ms_isolated=round(iso_ms, 4), launches_timed=len(kernel_events),
                algorithmic_flops_per_launch=fl, algorithmic_hbm_bytes_per_launch=ab,
                hbm_gbs_algorithmic=round(ab / (live_ms * 1e-3) / 1e9, 1), hbm_frac_of_peak=round(ab / (live_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))
    parity = parity_gate(w, mod, x) if want_parity else None
    del mod, x
    gc.collect()
    torch.cuda.empty_cache()
    return value, elapsed / args.steps * 1e3, roof, parity, w


def sharded_k65536(device, rank, world, steps=10):
    """BASELINE configs[4]: K=65536, D=512 sharded over the ranks; both exchange variants (RCCL MIN all-reduce of the
    packed-key planes, one-hop all-gather + MIN in the finalize), M = 8192 and M = 65536, each with a
    per-phase breakdown (search / exchange / finalize device time, host enqueue time per step) so that a measured
    scaling factor can be explained."""
    from vector_quantization.sharded import ShardedCodebookSearch

    K, D = 65536, 512
    g = torch.Generator().manual_seed(99)
    kl = K // world
    full = torch.randn((K, D), generator=g)
    shard = full[rank * kl:(rank + 1) * kl].to(device)
    table = full.to(device)  # replicated gather table (128 MiB)
    del full
    by_m = {}
    for M in (8192, 65536):
        x = torch.randn((M, D), generator=torch.Generator().manual_seed(1234)).to(device)
        res = {}
        for mode in ("all_reduce", "all_gather"):
            s = ShardedCodebookSearch(shard, full_codebook=table, reduction=mode)
            for _ in range(3):
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
            res[mode] = dict(value=round(M * steps / el, 1), ms_per_step=round(el / steps * 1e3, 3), idx_checksum=int(idx.sum().item()),
                             phases=s.profile_phases(x, steps=steps))
            if world == 1:
                break  # no exchange at N = 1: the two variants are the same launches
        best_mode = max(res, key=lambda m: res[m]["value"])
        by_m[M] = dict(value=res[best_mode]["value"], ms_per_step=res[best_mode]["ms_per_step"], reduction=best_mode, variants=res,
                       roofline_frac=round(2.0 * K * D * res[best_mode]["value"] / world / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
        del x
    head = by_m[8192]
    return dict(value=head["value"], unit="vectors/s", n_gpus=world, ms_per_step=head["ms_per_step"], reduction=head["reduction"],
                variants=head["variants"], roofline_frac=head["roofline_frac"], m65536=by_m[65536],
                config="K=65536 D=512 tokens replicated (M = 8192; m65536: M = 65536), codebook sharded K/N per GPU; per step ONE "
                       "search launch (K split into key planes, no init, no atomics), ONE exchange of the 8-byte packed "
                       "(distance, index) keys, ONE finalize launch (MIN over the candidate planes + gather from the replicated "
                       "table); M >= 32768: two halves, the first half's exchange under the second half's search")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)  # the chip needs ~30 ms of load to settle its clock
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--legs", default=None, help=f"comma list of extra workloads (N = 1 only; default {DEFAULT_LEGS} beside cfg2)")
    ap.add_argument("--no-legs", action="store_true")
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
    # rehearsal knobs (1-GPU box): all ranks on cuda:0 over gloo, to exercise the N > 1 code path
    one_device = os.environ.get("VQ_BENCH_ONE_DEVICE", "0") == "1"
    backend = os.environ.get("VQ_BENCH_BACKEND", "nccl")
    if world > 1:
        # the rendezvous comes FIRST: no HIP call (not even torch.cuda.is_available(), which initialises the runtime) has
        # been made in this process when the process group is created; the device is bound right after
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        tmo = datetime.timedelta(seconds=180)  # a collective that cannot complete fails the run instead of hanging it
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    if torch.cuda.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        dist.barrier(device_ids=[device.index] if backend == "nccl" else None)  # first collective: RCCL binds to this device

    from vector_quantization import native

    native.load()
    solo = rank == 0 and world == 1
    value, ms_per_step, roof, parity, w = run_workload(args.workload, args, device, rank, world, want_parity=solo and not args.no_cpu_baseline)

    legs = None
    leg_names = [] if (args.no_legs or world > 1) else [n for n in (args.legs if args.legs is not None else
                                                                    (DEFAULT_LEGS if args.workload == "cfg2" else "")).split(",") if n]
    if leg_names:
        legs = {}
        for n in leg_names:
            try:
                v, ms, r, par, wl = run_workload(n, args, device, rank, world, want_parity=not args.no_cpu_baseline)
                legs[n] = dict(value=round(v, 1), unit="vectors/s", ms_per_step=round(ms, 4), workload=wl["desc"],
                               rows_per_step=rows_per_step(wl), roofline=r, parity=par)
            except SystemExit:
                raise
            except Exception as e:  # a leg never takes the headline line down with it
                legs[n] = dict(error=f"{type(e).__name__}: {str(e)[:200]}")

    sharded = None
    if not args.no_sharded:
        try:
            with torch.no_grad():
                sharded = sharded_k65536(device, rank, world)
        except Exception as e:  # informational leg: never fail the bench line
            sharded = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    cpu = None
    if solo and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "vectors quantized/sec on [B*N,D]x[K,D] argmin; indices bit-exact vs CPU",
            "value": round(value, 1),
            "unit": "vectors/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
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
            "legs": legs,
            "sharded_k65536": sharded,
            "device": native.device_info(),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
