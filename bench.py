#!/usr/bin/env python3
"""bench.py — ICP iterations/sec at |F|=|M|=16384, |R|=256 (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one fixed-length registration pass of the hot path: starting from the identity transform,
ITERS_PER_STEP (=40, the length of the reference's profiling run, include/ICP/algorithms.hpp:2482-2494)
ICP iterations of the power-method / weighted pipeline on the synthetic kg-like pair (config 2 of
BASELINE.json), inputs resident in HBM, RBC already built.  One hipGraph launch per step.

N > 1 (launched by torch.distributed.run, one rank per GPU): a frame pair does not shard
 — "replicas only": every rank registers its own independent pair (seed + rank), no
data-path collective; value = iterations of all ranks / max-over-ranks time  ("scaling": "weak").

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (k_search): algorithmic bytes per launch (72 m + 32 |R| + 64) /
                its average launch-to-launch time, measured with HIP events on the engine's own stream.  In the
                default (fused, chained) form an iteration IS one k_search launch (it first turns the previous
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

M_POINTS, N_REPS, SIDE = 16384, 256, 128
ALPHA, SCALING = 2e2, 1e-6                       # src/ocl_icp_reg.cpp:88
ITERS_PER_STEP = 40
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_ITER = 72 * M_POINTS + 32 * N_REPS + 64


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


def cpu_baseline(F, M, fused, budget_s=12.0):
    """The oracle (CPU port of the same iteration) on this host's cores, bounded to ~budget_s."""
    from oracle import oracle as O
    cores = int(os.environ.get("ICP_BASELINE_THREADS", min(os.cpu_count() or 1, 16)))   # the search loops stop scaling at ~16 threads
    o = O.OracleICP(M_POINTS, N_REPS, ALPHA, SCALING, threads=cores, power_fast=True, fused=fused)
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
                      "%d threads, reductions serial" % (n, M_POINTS, N_REPS, el, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--power-mode", choices=["literal", "squared"], default="squared")
    ap.add_argument("--reduce-mode", choices=["reference", "fused"], default="fused")
    ap.add_argument("--batch", type=int, default=1,
                    help="independent registrations per GPU sharing each launch (BASELINE config 4 uses 64); default 1 = the headline metric")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

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
    F, M = icp_amd.synth_pair(SIDE, seed=0x1C9D5EED + rank)
    g = icp_amd.ICP(int(os.environ.get("ICP_BENCH_DEVICE", local_rank)))   # override: self-test of the N>1 path on a 1-GPU box
    g.init(M_POINTS, N_REPS, ALPHA, SCALING, batch=args.batch)
    g.setPowerMode(icp_amd.PowerMode.SQUARED if args.power_mode == "squared" else icp_amd.PowerMode.LITERAL)
    g.setReduceMode(icp_amd.ReduceMode.FUSED if args.reduce_mode == "fused" else icp_amd.ReduceMode.REFERENCE_ORDER)
    g.write(icp_amd.Memory.F, F)
    g.write(icp_amd.Memory.M, M)
    for bi in range(1, args.batch):                  # further independent pairs of this rank
        Fb, Mb = icp_amd.synth_pair(SIDE, seed=0x1C9D5EED + rank + 1000 * bi)
        g.write(icp_amd.Memory.F, Fb, batch_index=bi)
        g.write(icp_amd.Memory.M, Mb, batch_index=bi)
    g.buildRBC()
    g.sync()

    def barrier():
        if dist is not None:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        g.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        g.reset_transform()
        g.run_fixed(ITERS_PER_STEP)
    barrier()
    t0 = time.perf_counter()
    # the K steps; the engine brackets them with hipEvents on its own stream (roofline duration)
    ev_ms = g.time_run_fixed(ITERS_PER_STEP, args.steps, from_identity=True)
    barrier()
    elapsed = time.perf_counter() - t0

    total_t, total_iters = aggregate(dist, elapsed, args.steps * ITERS_PER_STEP * args.batch)

    # beside the metric (never part of `value`): latency of one whole registration = RBC construction + 40 iterations,
    # inputs resident, and the same with the two clouds uploaded from host memory first
    e2e = None
    if rank == 0 and args.batch == 1:
        reps = 20
        g.buildRBC(); g.sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            g.buildRBC(); g.reset_transform(); g.run_fixed(ITERS_PER_STEP)
        g.sync()
        resident_ms = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        for _ in range(reps):
            g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M)
            g.buildRBC(); g.reset_transform(); g.run_fixed(ITERS_PER_STEP)
        g.sync()
        upload_ms = (time.perf_counter() - t1) / reps * 1e3
        t1 = time.perf_counter()
        for _ in range(reps):
            g.buildRBC()
        g.sync()
        e2e = {"build_rbc_ms": (time.perf_counter() - t1) / reps * 1e3, "build_plus_%d_iterations_ms" % ITERS_PER_STEP: resident_ms,
               "with_upload_of_F_and_M_ms": upload_ms}

    # dominant kernel (k_search): average launch-to-launch time, HIP events on the engine's stream (rocprofv3's
    # per-dispatch average for the same kernel: profiles/r01_final_*_kernel_stats.csv).  Chained form: the timed
    # region itself is `steps` graphs of ITERS_PER_STEP k_search launches (+ one begin / end kernel per graph);
    # otherwise a graph holding only that kernel.
    fused = args.reduce_mode == "fused"
    launches = g.launches_per_iteration()
    names = (("search", 1), ("finalize", 8)) if fused else (("search", 1), ("means", 2), ("sij", 4), ("finalize", 8))
    kernel_us = {n + (" (separate launch)" if launches == 1 else ""): g.time_masked(mk, ITERS_PER_STEP, 20) for n, mk in names}
    iter_us = ev_ms * 1e3 / (args.steps * ITERS_PER_STEP)
    kernel_us["iteration (all kernels, from the timed region)"] = iter_us
    if launches == 1:
        search_us = iter_us
        kernel_name = "k_search<chained> (finalize of the previous iteration in its prologue)"
    else:
        search_us = kernel_us["search"]
        kernel_name = "k_search"
    achieved = ALGO_BYTES_PER_ITER / (search_us * 1e-6) / 1e9
    algo_flop = 18.0 * M_POINTS * (N_REPS + M_POINTS / N_REPS) + 100.0 * M_POINTS      # SURVEY.md §8d / BASELINE.md §3
    tflops = algo_flop / (search_us * 1e-6) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_search_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "ICP iterations/sec at |F|=|M|=16384, |R|=256",
            "value": total_iters / total_t,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": total_t / args.steps * 1e3,
            "us_per_iteration": total_t / (args.steps * ITERS_PER_STEP) * 1e6 / args.batch,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic kg-like pair, |F|=|M|=16384, |R|=256, power method, weighted, "
                                   "a=2e2 c=1e-6; step = %d fixed iterations (one hipGraph), RBC prebuilt" % ITERS_PER_STEP,
                       "parallelism": "replicas" if world > 1 else "single", "registrations_per_gpu": args.batch,
                       "power_start": args.power_mode,
                       "reduce_mode": args.reduce_mode, "launches_per_iteration": launches},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "launches_per_iteration": launches, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ITER, "avg_launch_us": search_us,
                         "kernel_us": kernel_us,
                         "valu_beside_it": {"algorithmic_flop_per_launch": algo_flop, "achieved_tflops": tflops,
                                            "peak_tflops_fp32_vector": 157.3, "frac": tflops / 157.3},
                         "note": "cache-resident at this size (1.19 MB/iteration): latency/VALU-bound, see DESIGN.md §5"},
        }
        if e2e is not None:
            line["registration_latency"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(F, M, fused)
        print(json.dumps(line))
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
