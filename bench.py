#!/usr/bin/env python3
"""bench.py — ICP iterations/sec at |F|=|M|=16384, |R|=256 (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config A|B|C] [--batch B]

A "step" is one fixed-length registration pass of the hot path: starting from the identity transform,
`iterations_per_step` (40 = the length of the reference's profiling run, include/ICP/algorithms.hpp:2482-2494; 10 at
config C) ICP iterations of the power-method / weighted pipeline on a synthetic pair, inputs resident in HBM, RBC
already built.  One hipGraph launch per step.

Workloads (icp_amd/workloads.py):  A = BASELINE configs[1] (|F|=|M|=16384, |R|=256; the headline, default),
B = configs[2] (65536 / 1024), C = configs[4] (2^20 / 4096); `--batch B` = B independent registrations sharing every
launch (configs[3] runs 64 per GPU).

N = 1 (default): batch 1 — the headline metric.  The line then also carries `other_configs`: the same measurement at
A x 64 registrations, B and C (fewer steps), each with its own algorithmic bytes / flops, HBM and fp32-VALU fractions,
launches per iteration and RBC construction time.
N > 1 (launched by torch.distributed.run, one rank per GPU): a frame pair does not shard — "replicas
only": every rank registers its own 64 independent pairs (config 4: seed base + 64 rank + i) unless --batch says otherwise,
no data-path collective; value = iterations of all ranks / max-over-ranks time ("scaling": "weak").  The single-GPU
figure of the same per-GPU work is `other_configs.A_x64` of the N = 1 line.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (k_search): algorithmic bytes per launch ((72 m + 32 |R| + 64) x registrations per launch,
                SURVEY.md §8d) / its average launch-to-launch time, measured with HIP events on the engine's own stream.
                In the default (fused, chained) form an iteration IS one k_search launch (it first turns the previous
                iteration's moments into T), so that time is the timed region / launches
  cpu_baseline  the CPU oracle ("port") timed on this host on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALPHA, SCALING = 2e2, 1e-6                       # src/ocl_icp_reg.cpp:88
ITERS_PER_STEP = 40
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3                         # fp32 vector peak (spec), same guide


def aggregate(dist, elapsed_s, units):
    """Whole-job numbers from per-rank ones: time = MAX over ranks, units = SUM over ranks.

    `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None:
        return elapsed_s, units
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    u = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def default_batch(world, requested):
    """Registrations per GPU: what --batch says; else 1 on one GPU (the headline), 64 per GPU on several (config 4)."""
    if requested:
        return requested
    return 1 if world == 1 else 64


def cpu_baseline(F, M, m, nr, fused, budget_s=12.0):
    """The oracle (CPU port of the same iteration) on this host's cores, bounded to ~budget_s."""
    from oracle import oracle as O
    cores = int(os.environ.get("ICP_BASELINE_THREADS", min(os.cpu_count() or 1, 16)))   # the search loops stop scaling at ~16 threads
    o = O.OracleICP(m, nr, ALPHA, SCALING, threads=cores, power_fast=True, fused=fused)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    o.step()                                     # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        if n % ITERS_PER_STEP == 0:
            o.write_t([0, 0, 0, 1, 0, 0, 0, 1])      # same workload as the GPU: fresh 40-iteration passes
        o.step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20000:
            break
    return {"value": n / el, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": "%d iterations of the same pair (|F|=|M|=%d, |R|=%d) in %.1f s; search loops OpenMP over "
                      "%d threads, reductions serial" % (n, m, nr, el, cores)}


def setup(icp_amd, device, cfg, batch, seed_index0, power_mode, reduce_mode):
    """An engine handle with `batch` registrations of workload `cfg` resident and the RBC built.
    Registration b uses pair seed_index0 + b of the config-4 family at A; at B and C the default pair (seeded by it)."""
    from icp_amd import workloads as W
    side, nr = W.CONFIGS[cfg]
    m = side * side
    g = icp_amd.ICP(device)
    g.init(m, nr, ALPHA, SCALING, batch=batch)
    g.setPowerMode(icp_amd.PowerMode.SQUARED if power_mode == "squared" else icp_amd.PowerMode.LITERAL)
    g.setReduceMode(icp_amd.ReduceMode.FUSED if reduce_mode == "fused" else icp_amd.ReduceMode.REFERENCE_ORDER)
    first = None
    for b in range(batch):
        if cfg == "A" and batch > 1:
            F, M = W.pair(icp_amd, seed_index0 + b)
        else:
            F, M = icp_amd.synth_pair(side, seed=W.BASE_SEED + seed_index0 + b)
        g.write(icp_amd.Memory.F, F, batch_index=b)
        g.write(icp_amd.Memory.M, M, batch_index=b)
        if first is None:
            first = (F, M)
    g.buildRBC()
    g.sync()
    return g, m, nr, first


def roofline_of(g, m, nr, batch, iters, steps, ev_ms, fused, traffic_key):
    """`roofline` object of the dominant kernel (k_search) for the timed region just measured (ev_ms = HIP-event time of
    `steps` graphs of `iters` iterations on the engine's stream)."""
    from icp_amd import workloads as W
    launches = g.launches_per_iteration()
    names = (("search", 1), ("finalize", 8)) if fused else (("search", 1), ("means", 2), ("sij", 4), ("finalize", 8))
    reps = 20 if m <= 65536 else 2
    kernel_us = {n + (" (separate launch)" if launches == 1 else ""): g.time_masked(mk, iters, reps) for n, mk in names}
    iter_us = ev_ms * 1e3 / (steps * iters)
    kernel_us["iteration (all kernels, from the timed region)"] = iter_us
    if launches == 1:
        search_us = iter_us
        kernel_name = "k_search<chained> (finalize of the previous iteration in its prologue)"
    else:
        # the search's share of the timed iterations: the iteration minus the other kernels (each timed alone, boundary
        # included).  The search timed alone repeats ONE state — the converged one, where the pruning is at its best — and
        # would understate a run whose first iterations start far from it (C: 248 against 270 us); it stays in kernel_us.
        search_us = iter_us - sum(v for k, v in kernel_us.items() if not k.startswith(("search", "iteration")))
        kernel_us["search (in the timed iterations: iteration - the other kernels)"] = search_us
        kernel_name = "k_search"
    bytes_launch = W.algorithmic_bytes(m, nr) * batch          # one launch serves every registration of the batch
    flop_launch = W.algorithmic_flop(m, nr) * batch
    achieved = bytes_launch / (search_us * 1e-6) / 1e9
    tflops = flop_launch / (search_us * 1e-6) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(traffic_key)
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": kernel_name, "launches_per_iteration": launches, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_launch, "registrations_per_launch": batch, "avg_launch_us": search_us,
            "kernel_us": kernel_us,
            "valu_beside_it": {"algorithmic_flop_per_launch": flop_launch, "achieved_tflops": tflops,
                               "peak_tflops_fp32_vector": VALU_PEAK_TFLOPS, "frac": tflops / VALU_PEAK_TFLOPS},
            "note": ("algorithmic bytes count the 8 m correspondence write of every iteration; fused graphs of a fixed length store "
                     "it in their last iteration only (DESIGN.md §5). " +
                     ("Cache-resident at this size: latency / VALU-bound, see DESIGN.md §5" if m * batch <= (1 << 21) else
                      "fp32-VALU-bound stage 1 (pruned brute force over the representatives), see DESIGN.md §5"))}


def measure_config(icp_amd, device, cfg, batch, steps, warmup, iters, power_mode="squared", reduce_mode="fused"):
    """One entry of `other_configs`: the same step / timing as the headline at another workload (single process)."""
    from icp_amd import workloads as W
    g, m, nr, _ = setup(icp_amd, device, cfg, batch, 0, power_mode, reduce_mode)
    for _ in range(warmup):
        g.run_fixed_fresh(iters)
    g.sync()
    t0 = time.perf_counter()
    ev_ms = g.time_run_fixed(iters, steps, from_identity=True)
    g.sync()
    wall = time.perf_counter() - t0
    rl = roofline_of(g, m, nr, batch, iters, steps, ev_ms, reduce_mode == "fused",
                     "k_search_hbm_bytes_per_launch_%s" % (cfg if batch == 1 else "%s_x%d" % (cfg, batch)))
    nb = 5 if m > 65536 else 20                      # the RBC construction, back to back (cached graph, warm like the headline's)
    g.buildRBC()
    g.sync()
    t1 = time.perf_counter()
    for _ in range(nb):
        g.buildRBC()
    g.sync()
    build_ms = (time.perf_counter() - t1) / nb * 1e3
    g.close()
    total_iters = steps * iters * batch
    return {"workload": "|F|=|M|=%d, |R|=%d, %d registration(s) per launch" % (m, nr, batch), "steps": steps, "warmup": warmup,
            "iterations_per_step": iters, "iterations_per_s": total_iters / wall,
            "us_per_iteration": wall / total_iters * 1e6, "us_per_batched_iteration": wall / (steps * iters) * 1e6,
            "launches_per_iteration": rl["launches_per_iteration"], "build_rbc_ms": build_ms,
            "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"],
            "algorithmic_flop_per_launch": rl["valu_beside_it"]["algorithmic_flop_per_launch"],
            "k_search_avg_launch_us": rl["avg_launch_us"], "kernel_us": rl["kernel_us"],
            "hbm_gbs": rl["achieved"], "hbm_frac": rl["frac"], "hbm_traffic_bytes_per_launch": rl["traffic"],
            "valu_tflops": rl["valu_beside_it"]["achieved_tflops"], "valu_frac": rl["valu_beside_it"]["frac"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the A x 64 / B / C measurements of the N = 1 line")
    ap.add_argument("--power-mode", choices=["literal", "squared"], default="squared")
    ap.add_argument("--reduce-mode", choices=["reference", "fused"], default="fused")
    ap.add_argument("--config", choices=["A", "B", "C"], default="A",
                    help="workload of the line: A = BASELINE configs[1] (default, the metric), B = configs[2], C = configs[4]")
    ap.add_argument("--batch", type=int, default=0,
                    help="independent registrations per GPU sharing each launch; default: 1 on one GPU (the headline metric), "
                         "64 per GPU on several (BASELINE config 4)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    batch = default_batch(world, args.batch)
    iters = 10 if args.config == "C" else ITERS_PER_STEP
    steps = args.steps if args.steps is not None else (200 if args.config == "A" else 40 if args.config == "B" else 5)
    warmup = args.warmup if args.warmup is not None else (20 if args.config == "A" else 5 if args.config == "B" else 1)

    # torch first (when present) so that the process holds ONE HIP runtime: libicp_amd.so then binds
    # to the libamdhip64.so.7 torch has already loaded.  torch is plumbing here (barrier / reductions).
    dist = None
    torch = None
    try:
        import torch  # noqa: F401
    except Exception:
        torch = None
    if world > 1:
        if torch is None:
            raise SystemExit("bench.py: torch.distributed is required for --gpus > 1")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ICP_BENCH_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")   # nccl == RCCL
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("ICP_BENCH_DEVICE", local_rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import icp_amd
    from icp_amd import workloads as W
    device = int(os.environ.get("ICP_BENCH_DEVICE", local_rank))   # override: self-test of the N>1 path on a 1-GPU box
    g, m, nr, (F, M) = setup(icp_amd, device, args.config, batch, rank * batch, args.power_mode, args.reduce_mode)

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        g.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(warmup):
        g.run_fixed_fresh(iters)                     # a fresh registration: from the identity transform, one graph
    barrier()
    t0 = time.perf_counter()
    # the K steps; the engine brackets them with hipEvents on its own stream (roofline duration)
    ev_ms = g.time_run_fixed(iters, steps, from_identity=True)
    barrier()
    elapsed = time.perf_counter() - t0

    total_t, total_iters = aggregate(dist, elapsed, steps * iters * batch)

    # beside the metric (never part of `value`): latency of one whole registration = RBC construction + the iterations,
    # inputs resident, and the same with the two clouds uploaded from host memory first
    e2e = None
    if rank == 0 and batch == 1:
        reps = 20 if m <= 65536 else 3
        g.buildRBC(); g.sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            g.buildRBC(); g.run_fixed_fresh(iters)
        g.sync()
        resident_ms = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        for _ in range(reps):
            g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M)
            g.buildRBC(); g.run_fixed_fresh(iters)
        g.sync()
        upload_ms = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        for _ in range(reps):
            g.buildRBC()
        g.sync()
        e2e = {"build_rbc_ms": (time.perf_counter() - t1) / reps * 1e3, "build_plus_%d_iterations_ms" % iters: resident_ms,
               "with_upload_of_F_and_M_ms": upload_ms}

    # dominant kernel (k_search): average launch-to-launch time, HIP events on the engine's stream (rocprofv3's
    # per-dispatch average for the same kernel: profiles/).  Chained form: the timed region itself is `steps` graphs of
    # `iters` k_search launches (+ one reset / end kernel per graph); otherwise a graph holding only that kernel.
    fused = args.reduce_mode == "fused"
    tkey = "k_search_hbm_bytes_per_launch" if (args.config == "A" and batch == 1) else \
           "k_search_hbm_bytes_per_launch_%s" % (args.config if batch == 1 else "%s_x%d" % (args.config, batch))
    roofline = roofline_of(g, m, nr, batch, iters, steps, ev_ms, fused, tkey) if rank == 0 else None
    launches = g.launches_per_iteration()
    g.close()

    if rank == 0:
        cfg_text = {"A": "configs[1]: synthetic kg-like pair, |F|=|M|=16384, |R|=256",
                    "B": "configs[2]: synthetic VGA RGB-D cloud subsampled to |F|=|M|=65536, |R|=1024",
                    "C": "configs[4]: single registration |F|=|M|=2^20, |R|=4096"}[args.config]
        if args.config == "A" and batch > 1:
            cfg_text = "configs[3]: independent frame pairs of |F|=|M|=16384, |R|=256, %d per GPU sharing each launch, no RCCL" % batch
        line = {
            "metric": "ICP iterations/sec at |F|=|M|=%d, |R|=%d" % (m, nr),
            "value": total_iters / total_t,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": total_t / steps * 1e3,
            "us_per_iteration": total_t / (steps * iters) * 1e6 / batch,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg_text + ", power method, weighted, a=2e2 c=1e-6; step = %d fixed iterations (one hipGraph), RBC prebuilt" % iters,
                       "parallelism": "replicas" if world > 1 else "single", "registrations_per_gpu": batch,
                       "power_start": args.power_mode,
                       "reduce_mode": args.reduce_mode, "launches_per_iteration": launches},
            "roofline": roofline,
        }
        if world > 1:
            line["config"]["scaling_reference"] = ("per-GPU work is %d registrations per launch: the single-GPU figure of the same "
                                                   "work is other_configs.A_x64 of the N = 1 line, not its batch-1 value" % batch)
        if e2e is not None:
            line["registration_latency"] = e2e
        if world == 1 and args.config == "A" and batch == 1 and not args.no_other_configs:
            # the other BASELINE configs on the same GPU, same step definition, fewer steps (C: 3 steps of 10 iterations)
            oc = {}
            for key, cfg, b, st, wu, it in (("A_x64", "A", 64, 20, 3, ITERS_PER_STEP), ("B", "B", 1, 40, 5, ITERS_PER_STEP), ("C", "C", 1, 3, 1, 10)):
                oc[key] = measure_config(icp_amd, device, cfg, b, st, wu, it, args.power_mode, args.reduce_mode)
            line["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline and m <= 65536:
            line["cpu_baseline"] = cpu_baseline(F, M, m, nr, fused)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
