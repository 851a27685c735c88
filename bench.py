#!/usr/bin/env python3
"""bench.py — ICP iterations/sec at |F|=|M|=16384, |R|=256 (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--config A|B|C] [--batch B]

A "step" is one fixed-length registration pass of the hot path: starting from the identity transform,
`iterations_per_step` (40 = the length of the reference's profiling run, include/ICP/algorithms.hpp:2482-2494; 10 at
config C) ICP iterations of the power-method / weighted pipeline on a synthetic pair, inputs resident in HBM, RBC
already built.  One hipGraph launch per step.  Every pass is a FRESH registration: the first search of a
pass is seeded from the queries' own grid cells, never from the previous pass's answer (`warm_seed_us_per_iteration` shows
what that seed would have been worth).

Workloads (icp_amd/workloads.py):  A = BASELINE configs[1] (|F|=|M|=16384, |R|=256; the headline, default),
B = configs[2] (65536 / 1024), C = configs[4] (2^20 / 4096); `--batch B` = B independent registrations sharing every
launch (configs[3] runs 64 per GPU).

--gpus 1 (default): batch 1 — the headline metric.  The line also carries `other_configs`: the same measurement at
A x 64 registrations, B and C (fewer steps) and frame-to-frame tracking.
--gpus N > 1: a frame pair does not shard — "replicas only": every GPU registers its own 64 independent
pairs (config 4) unless --batch says otherwise, no data-path collective, no RCCL.  Two ways to get there, same numbers:
  * launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's form): one rank per
    GPU; torch.distributed supplies only the barrier and the max / sum / gather of the per-rank numbers;
  * plain `python bench.py --gpus N`: devices 0..N-1 are driven in-process through the library's own icp_batch_* (one host
    thread + one stream per device).
`--gpus` must equal the ranks / devices actually used, or the run fails.  value = iterations of all GPUs / max-over-GPUs
time ("scaling": "weak").  EVERY line carries `config4_per_gpu_value` — iterations/s of one GPU at config 4's per-GPU share
(64 registrations per launch) — so that N = 1 -> 8 reads off ONE field; the N = 1 `value` itself is the batch-1 headline.

Prints ONE JSON line (rank 0) as the LAST line of stdout, shorter than 4 KB (`compact_line`): the contract's fields, `roofline`,
`cpu_baseline` and flat scalars for the other BASELINE configs.  Everything else that is measured (per-kernel times, tracking
distributions, the invalid-point cases, the mode comparison, the thread sweep) goes to bench_extra.json beside this file (stderr gets one
line that says so; ICP_BENCH_STDERR_RECORD=1: the whole record).
Objects of the line:
  roofline      dominant kernel (k_search): algorithmic bytes per launch ((72 m + 32 |R| + 64) x registrations per launch,
                SURVEY.md §8d) / its average launch-to-launch time, measured with HIP events on the engine's own stream.
                In the default (fused, chained) form an iteration IS one k_search launch (it first turns the previous
                iteration's moments into T), so that time is the timed region / launches
  cpu_baseline  the CPU oracle ("port") timed on this host on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALPHA, SCALING = 2e2, 1e-6                       # src/ocl_icp_reg.cpp:88
ITERS_PER_STEP = 40
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3                         # fp32 vector peak (spec), same guide
SIMDS, CLOCK_HZ = 1024, 2.4e9                    # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles


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


def gather_per_rank(dist, value):
    """[value of rank 0, value of rank 1, ..] on every rank (a report field; nothing on the data path)."""
    if dist is None:
        return [float(value)]
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(x.item()) for x in out]


def default_batch(world, requested):
    """Registrations per GPU: what --batch says; else 1 on one GPU (the headline), 64 per GPU on several (config 4)."""
    if requested:
        return requested
    return 1 if world == 1 else 64


def resolve_launch(gpus, env):
    """How this invocation spans `gpus` GPUs: ("single", 1), ("ranks", world) under torch.distributed.run (WORLD_SIZE set by
    the launcher; it must equal --gpus) or ("inprocess", gpus) for a plain `python bench.py --gpus N` (icp_batch_* over the
    devices 0..N-1).  Raises SystemExit when --gpus and the launch disagree."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        world = int(ws)
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to print a line "
                             "whose n_gpus would not be the GPUs used" % (gpus, world))
        return ("ranks", world) if world > 1 else ("single", 1)
    return ("inprocess", gpus) if gpus > 1 else ("single", 1)


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def cpu_share():
    """The cores this process may actually use: the scheduler affinity and the cgroup's CPU quota (a one-GPU box of the pool has a share of
    16 of its host's 256 cores — thread counts above the share time-slice, which is what the tail of the sweep shows)."""
    share = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                share = min(share, float(quota) / period)
            break
        except Exception:
            continue
    return share


def sweep_counts(share, host):
    """Thread counts of the CPU baseline's sweep: 1, 8, 16, 64 and the process's CPU share — never more threads than the share (a
    one-GPU box has 16 of its host's 256 cores: 256 threads on them time-slice and measure nothing but the scheduler)."""
    cap = max(1, min(int(host), int(share)))
    return sorted({t for t in (1, 8, 16, 64, cap) if 1 <= t <= cap})


def cpu_baseline(F, M, m, nr, fused, budget_s=12.0):
    """The oracle (CPU port of the same iteration) on this host's cores: a thread sweep {1, 8, 16, 64, share} up to the process's CPU share, each count timed on
    a bounded sample of the same workload (fresh 40-iteration passes of the benchmark pair), the best one reported as `value` with the
    sweep beside it.  OpenMP over the queries (search, transform, weights) and over the 64-pair blocks of the moment reduction —
    deterministic per-thread partials: the same bits as one thread (tests/test_oracle_golden.py)."""
    from oracle import oracle as O
    host = os.cpu_count() or 1
    env = os.environ.get("ICP_BASELINE_THREADS")
    share = cpu_share()
    counts = [int(env)] if env else sweep_counts(share, host)
    o = O.OracleICP(m, nr, ALPHA, SCALING, threads=counts[0], power_fast=True, fused=fused)
    o.write_f(F)
    o.write_m(M)
    o.build_rbc()
    per = budget_s / len(counts)
    sweep, total_n, total_t = [], 0, 0.0
    for t in counts:
        o.L.orc_icp_set_threads(o.h, t)
        o.step()                                     # warm-up (the thread team of this size)
        n, t0 = 0, time.perf_counter()
        while True:
            if n % ITERS_PER_STEP == 0:
                o.write_t([0, 0, 0, 1, 0, 0, 0, 1])      # same workload as the GPU: fresh 40-iteration passes
            o.step()
            n += 1
            el = time.perf_counter() - t0
            if el >= per or n >= 20000:
                break
        sweep.append({"threads": t, "iterations_per_s": n / el, "iterations": n, "seconds": el})
        total_n += n
        total_t += el
    best = max(sweep, key=lambda x: x["iterations_per_s"])
    return {"value": best["iterations_per_s"], "unit": "iterations/s", "cores": best["threads"], "threads": best["threads"], "host_cores": host,
            "cpu_share": share, "kind": "port", "sweep": sweep,
            "sample_short": "%d iterations of the same pair in %.1f s, OpenMP thread sweep %s, best count reported" % (total_n, total_t, [x["threads"] for x in sweep]),
            "sample": "%d iterations of the same pair (|F|=|M|=%d, |R|=%d) in %.1f s over a sweep of %s threads of the host's %d cores (this process's share: %.4g) "
                      "(~%.0f s each); `value` = the best count; search, transform, weights and the block partials of the moment "
                      "reduction in OpenMP, deterministic per-thread partials" % (total_n, m, nr, total_t, [x["threads"] for x in sweep], host, share, per)}


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
        F, M = pair_of(icp_amd, cfg, batch, seed_index0 + b)
        g.write(icp_amd.Memory.F, F, batch_index=b)
        g.write(icp_amd.Memory.M, M, batch_index=b)
        if first is None:
            first = (F, M)
    g.buildRBC()
    g.sync()
    return g, m, nr, first


SETUP_MS = float(os.environ.get("ICP_BENCH_SETUP_MS", "25"))


def settle(g, iters):
    """Part of the untimed set-up, before the W warm-up steps: the graph of a step is captured and instantiated, and the device is kept
    busy with it for SETUP_MS milliseconds.  An MI355X that has been idle runs its first ~10 ms of work below its steady clocks
    (20 passes right after 5: 9.3 us per iteration; after another 20: 8.95; tools/diag/overhead.py) — a registration service is
    never in that state, a freshly started benchmark process always is.  Returns the number of passes run."""
    n, t0 = 0, time.perf_counter()
    while True:
        g.run_fixed_fresh(iters)
        g.sync()
        n += 1
        if (time.perf_counter() - t0) * 1e3 >= SETUP_MS or n >= 10000:
            return n


def pair_of(icp_amd, cfg, batch, index):
    from icp_amd import workloads as W
    if cfg == "A" and batch > 1:
        return W.pair(icp_amd, index)
    return icp_amd.synth_pair(W.CONFIGS[cfg][0], seed=W.BASE_SEED + index)


def traffic_of(key):
    """Fabric-side bytes per launch of the dominant kernel, from the committed counter passes (profiles/traffic.json): NOT
    measured by this run — PMC counters need rocprofv3 around the process — and labelled so."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None, None
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None, None
    cfgkey = {"k_search_hbm_bytes_per_launch": "detail_A", "k_search_hbm_bytes_per_launch_B": "detail_B",
              "k_search_hbm_bytes_per_launch_C": "detail_C", "k_search_hbm_bytes_per_launch_A_x64": "detail_Ax64"}.get(key)
    det = t.get(cfgkey) if cfgkey else None
    src = {"file": "profiles/traffic.json", "summary": t.get("_source"), "measured_at_commit": t.get("_commit"),
           "measured_by_this_run": False,
           "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes around this bench command, corrected per MI355X_MICROARCH.md"}
    issue = None
    if det and det.get("SQ_INSTS_VALU") and det.get("avg_dispatch_us_kernel_trace"):
        # executed work, not algorithmic: the kernel's wave-level VALU instructions (SQ_INSTS_VALU) x the MEASURED issue cost of its instruction
        # mix (profiles/valu_issue.json: tests/cpp/valu_issue_probe.hip's SIMD cycles per wave64 instruction by class — 2.2 for plain f32 / int
        # add, mul, fma, mov at 8 waves per SIMD, 4.1 - 4.3 for packed f32, min / max, compares, DPP, shifts, f64 — weighted by the classes'
        # shares in the loops of the kernel's ISA, tools/diag/valu_mix.py) / (1024 SIMDs x the cycles of the dispatch: SQ_BUSY_CYCLES / 32
        # shader engines of the same profile where it holds them, else its duration x 2.4 GHz)
        cyc, mix_src = 4.0, "assumed 4 cycles per instruction (profiles/valu_issue.json not found)"
        try:
            km = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))["_kernel_mix"][cfgkey[len("detail_"):]]
            cyc, mix_src = float(km["cycles_per_valu_instruction_from_the_mix"]), "profiles/valu_issue.json (_kernel_mix, %d waves per SIMD)" % km["waves_per_simd"]
        except Exception:
            pass
        # (the dispatch's cycles: SQ_BUSY_CYCLES / 32 shader engines — at |F| = 2^20 it says 2.15 GHz, the clock a saturated vector unit runs at —
        # but never more than the duration at 2.4 GHz: for a 9 us dispatch the counter's window is wider than the kernel)
        cycles = det["avg_dispatch_us_kernel_trace"] * 1e-6 * CLOCK_HZ
        if det.get("SQ_BUSY_CYCLES"):
            cycles = min(cycles, det["SQ_BUSY_CYCLES"] / 32.0)
        issue = {"valu_issue_frac": det["SQ_INSTS_VALU"] * cyc / SIMDS / cycles,
                 "valu_issue_frac_if_every_instruction_cost_4_cycles": det["SQ_INSTS_VALU"] * 4.0 / SIMDS / cycles,
                 "valu_issue_frac_if_every_instruction_were_full_rate": det["SQ_INSTS_VALU"] * 2.24 / SIMDS / cycles,
                 "cycles_per_valu_instruction": cyc, "cycles_source": mix_src, "dispatch_cycles_per_simd": cycles,
                 "dispatch_cycles_source": "min (SQ_BUSY_CYCLES / 32, duration x 2.4 GHz)" if det.get("SQ_BUSY_CYCLES") else "duration x 2.4 GHz",
                 "SQ_INSTS_VALU_per_launch": det["SQ_INSTS_VALU"], "kernel_us_in_that_profile": det["avg_dispatch_us_kernel_trace"],
                 "source": src["file"], "measured_at_commit": src["measured_at_commit"], "measured_by_this_run": False}
    return t.get(key), src, issue


def roofline_of(g, m, nr, batch, iters, steps, ev_ms, fused, traffic_key):
    """`roofline` object of the dominant kernel (k_search) for the timed region just measured (ev_ms = HIP-event time of
    `steps` graphs of `iters` iterations on the engine's stream: the steps the events bracket, all but the first of the region)."""
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
    # what the timed graph really moves: a fused graph of a fixed length stores the 8 m correspondence bytes in its LAST iteration only
    # (nothing in between can read them) — the §8d figure counts them every iteration
    moved_launch = bytes_launch - (8 * m * batch * (iters - 1) / iters if fused else 0)
    flop_launch = W.algorithmic_flop(m, nr) * batch
    achieved = bytes_launch / (search_us * 1e-6) / 1e9
    tflops = flop_launch / (search_us * 1e-6) / 1e12
    traffic, tsrc, issue = traffic_of(traffic_key)
    return {"bound": "hbm", "kernel": kernel_name, "launches_per_iteration": launches, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_bytes_per_launch": bytes_launch, "bytes_moved_per_launch": moved_launch,
            "frac_moved": moved_launch / (search_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "registrations_per_launch": batch, "avg_launch_us": search_us,
            "kernel_us": kernel_us,
            "valu_beside_it": {"algorithmic_flop_per_launch": flop_launch, "bruteforce_equivalent_tflops": tflops,
                               "peak_tflops_fp32_vector": VALU_PEAK_TFLOPS,
                               "note": "brute-force-equivalent rate (18 flop per distance of an exhaustive Q x R + list scan): the dense "
                                       "kernels prune, so this is NOT a utilisation; executed work is `executed`",
                               "executed": issue},
            "note": ("`frac` prices SURVEY.md §8d's algorithmic bytes (the 8 m correspondence write counted every iteration); `frac_moved` what the "
                     "timed graph stores. " +
                     ("Cache-resident at this size: latency / VALU-bound, see DESIGN.md §5" if m * batch <= (1 << 21) else
                      "fp32-VALU-bound stage 1 (pruned brute force over the representatives), see DESIGN.md §5"))}


def measure_config(icp_amd, device, cfg, batch, steps, warmup, iters, power_mode="squared", reduce_mode="fused", warm_seed=True):
    """One entry of `other_configs`: the same step / timing as the headline at another workload (single process)."""
    g, m, nr, _ = setup(icp_amd, device, cfg, batch, 0, power_mode, reduce_mode)
    settle(g, iters)
    for _ in range(warmup):
        g.run_fixed_fresh(iters)
    g.sync()
    t0 = time.perf_counter()
    ev_ms, ev_steps = g.time_run_fixed_tail(iters, steps, from_identity=True)
    g.sync()
    wall = time.perf_counter() - t0
    rl = roofline_of(g, m, nr, batch, iters, ev_steps, ev_ms, reduce_mode == "fused",
                     "k_search_hbm_bytes_per_launch_%s" % (cfg if batch == 1 else "%s_x%d" % (cfg, batch)))
    nb = 5 if m > 65536 else 20                      # the RBC construction, back to back (cached graph, warm like the headline's)
    g.buildRBC()
    g.sync()
    t1 = time.perf_counter()
    for _ in range(nb):
        g.buildRBC()
    g.sync()
    build_ms = (time.perf_counter() - t1) / nb * 1e3
    dense = bool(g.search_layout()[0])               # only the dense search prunes stage 1 with a seed: only there does a pass's first seed matter
    g.close()
    warm_us = None
    if warm_seed and dense:
        # the same passes with the first search of every pass seeded by the PREVIOUS pass's (converged) nearest representatives —
        # what re-registering one pair gave before round 3; reported beside the figure, never as the figure
        os.environ["ICP_AMD_WARM_SEED"] = "1"
        try:
            w, _, _, _ = setup(icp_amd, device, cfg, batch, 0, power_mode, reduce_mode)
            settle(w, iters)
            for _ in range(warmup):
                w.run_fixed_fresh(iters)
            w.sync()
            wms, wst = w.time_run_fixed_tail(iters, steps, from_identity=True)
            warm_us = wms * 1e3 / (wst * iters * batch)
            w.close()
        finally:
            del os.environ["ICP_AMD_WARM_SEED"]
    total_iters = steps * iters * batch
    ex = rl["valu_beside_it"]["executed"]
    return {"workload": "|F|=|M|=%d, |R|=%d, %d registration(s) per launch" % (m, nr, batch), "steps": steps, "warmup": warmup,
            "iterations_per_step": iters, "iterations_per_s": total_iters / wall,
            "us_per_iteration": wall / total_iters * 1e6, "us_per_batched_iteration": wall / (steps * iters) * 1e6,
            "seed_of_each_pass": "fresh (queries' own grid cells)", "warm_seed_us_per_iteration": warm_us,
            "launches_per_iteration": rl["launches_per_iteration"], "build_rbc_ms": build_ms,
            "algorithmic_bytes_per_launch": rl["algorithmic_bytes_per_launch"], "bytes_moved_per_launch": rl["bytes_moved_per_launch"],
            "algorithmic_flop_per_launch": rl["valu_beside_it"]["algorithmic_flop_per_launch"],
            "k_search_avg_launch_us": rl["avg_launch_us"], "kernel_us": rl["kernel_us"],
            "hbm_gbs": rl["achieved"], "hbm_frac": rl["frac"], "hbm_frac_moved": rl["frac_moved"],
            "hbm_traffic_bytes_per_launch": rl["traffic"], "hbm_traffic_source": rl["traffic_source"],
            "bruteforce_equivalent_tflops": rl["valu_beside_it"]["bruteforce_equivalent_tflops"],
            "valu_issue_frac": ex["valu_issue_frac"] if ex else None, "valu_issue_source": ex}


def measure_modes(icp_amd, device, g_default, power_mode, reduce_mode):
    """What the benchmarked (fused + squared) mode is, in numbers, against the reference-order / literal pipeline on the same
    pair: that pipeline's own time per iteration, and the free-running difference of the two registrations."""
    import numpy as np
    r, m, nr, _ = setup(icp_amd, device, "A", 1, 0, "literal", "reference")
    settle(r, ITERS_PER_STEP)
    for _ in range(5):
        r.run_fixed_fresh(ITERS_PER_STEP)
    r.sync()
    rms, rst = r.time_run_fixed_tail(ITERS_PER_STEP, 50, from_identity=True)
    ref_us = rms * 1e3 / (rst * ITERS_PER_STEP)
    launches = r.launches_per_iteration()
    out = {"reference_order_us_per_iteration": ref_us, "reference_order_launches_per_iteration": launches}
    # the same reductions with the squared power start: what of the reference-order mode's time is the reduction trees (this figure
    # minus the benchmarked one) and what the literal power loop (the figure above minus this one)
    try:
        q, _, _, _ = setup(icp_amd, device, "A", 1, 0, "squared", "reference")
        settle(q, ITERS_PER_STEP)
        for _ in range(5):
            q.run_fixed_fresh(ITERS_PER_STEP)
        q.sync()
        qms, qst = q.time_run_fixed_tail(ITERS_PER_STEP, 50, from_identity=True)
        out["reference_order_squared_us_per_iteration"] = qms * 1e3 / (qst * ITERS_PER_STEP)
        q.close()
    except Exception as e:                           # noqa: BLE001
        out["reference_order_squared_us_per_iteration"] = None
        out["reference_order_squared_error"] = "%s: %s" % (type(e).__name__, e)
    Mem = icp_amd.Memory
    res = []
    for h in (g_default, r):
        h.reset_transform()
        h.buildRBC()
        k = h.run()
        res.append((k, h.read(Mem.T), h.read(Mem.NN_ID)["id"]))
    (kf, Tf, idf), (kr, Tr, idr) = res
    # Both modes under a float64 solution (tests/float64_ref.py — measurement infrastructure, numpy on the host): the same iterations
    # restated in float64, fed the correspondences the ENGINE found in each iteration (single steps: the same bits as run ())
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import float64_ref as R64
    F, M = pair_of(icp_amd, "A", 1, 0)
    scene = float(np.abs(F[:, :3]).max())
    vs, T64 = {}, []
    for name, h, k, Tend in (("benchmarked", g_default, kf, Tf), ("reference_order", r, kr, Tr)):
        h.reset_transform()
        h.buildRBC()
        f = R64.Float64ICP(F, M, ALPHA, SCALING)
        for _ in range(int(k)):
            h.step()
            f.step(h.read(Mem.NN_ID)["id"])
        same = bool(np.array_equal(h.read(Mem.T).view(np.uint32), Tend.view(np.uint32)))
        vs[name] = dict(R64.errors_against(Tend, f.T, scene), steps_equal_run_bit_for_bit=same)
        T64.append(f.T)
    between32, between64 = R64.errors_against(Tf, Tr, scene), R64.errors_against(T64[0], T64[1], scene)
    out["mode_note"] = {
        "benchmarked": "%s reductions + %s power start (the handle's defaults)" % (reduce_mode, power_mode),
        "against": "reference-order reductions + literal power method (ICP_AMD_MODE=reference), same pair, both run to convergence",
        "k": [int(kf), int(kr)], "max_abs_dq": float(np.abs(Tf[:4] - Tr[:4]).max()),
        "max_abs_dt_mm": float(np.abs(Tf[4:7] - Tr[4:7]).max()), "abs_ds": float(abs(Tf[7] - Tr[7])),
        "norm_t_mm": float(np.linalg.norm(Tr[4:7])), "scene_scale_mm": scene, "identical_ids_frac": float(np.mean(idf == idr)),
        "differing_ids": int((idf != idr).sum()),
        "vs_float64": {
            "what": "each mode's final [q | t, s] against the float64 restatement of its own iterations (its own correspondences): dq = |q - q64|, "
                    "dt_over_t = |t - t64| / |t64| (the strict norm), dt_over_scene = |t - t64| / largest landmark coordinate, ds_over_s",
            "benchmarked": vs["benchmarked"], "reference_order": vs["reference_order"],
            "between_the_modes_fp32": between32, "between_their_float64_solutions": between64,
            "reading": "every mode is within 1e-5 of its own float64 solution in every component in the strict norm, the benchmarked one closer; "
                       "the two modes differ by what the float64 solutions of their correspondence sets differ by (a near-tie correspondence "
                       "or two of 16384): no arithmetic of the reductions can bring two free-running registrations closer than that"},
        "parity": "each mode equals its own oracle restatement bit for bit (tests/test_gpu_parity.py); for the same T both modes give the "
                  "same correspondences bit for bit (test_teacher_forced_default_modes_at_A); tests/test_float64_contract.py and "
                  "test_fused_run_and_cross_mode_tolerance assert the float64 statement"}
    r.close()
    return out


def _dist(a):
    import numpy as np
    a = np.asarray(a, float)
    return {"p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)), "p99": float(np.percentile(a, 99)), "max": float(a.max()),
            "mean": float(a.mean())}


def _track_report(hops, elapsed, gaps_us, lat_us, ks, launches=None, period=0):
    """One tracking variant: frames/s and the distributions of the SAME pass.  `gap` = time between two results reaching the host,
    `latency` = submit call -> result on the host.  A frame's time follows its iteration count k (a warm start on a sequence that
    reverses direction needs more iterations at the turning points): `gap_over_same_k` is every frame's gap over the median gap of the
    frames with the same k — what is left is jitter, not workload.  With frames in flight a frame's gap also depends on its
    neighbours (a 3-iteration frame behind a 40-iteration one is done the moment that one is; behind another 3-iteration frame it waits
    for the host's next submit): `gap_over_same_hop` compares every frame with the frames at the same position of the sequence's period
    (same k, same neighbours) — the jitter figure for the warm-start passes."""
    import numpy as np
    gaps, ks = np.asarray(gaps_us, float), np.asarray(ks)
    ratio = []
    for k in np.unique(ks):
        sel = gaps[ks == k]
        if len(sel) >= 4:
            ratio.extend(sel / np.median(sel))
    hop_ratio = []
    if period:
        for j in range(period):
            sel = gaps[j::period]
            if len(sel) >= 4:
                hop_ratio.extend(sel / np.median(sel))
    out = {"frames": hops, "frames_per_s": hops / elapsed, "ms_per_frame": elapsed / hops * 1e3,
           "completion_gap_us": _dist(gaps), "latency_us": _dist(lat_us),
           "iterations": {"mean": float(ks.mean()), "p50": float(np.percentile(ks, 50)), "max": int(ks.max())},
           "gap_over_same_k": ({"p99": float(np.percentile(ratio, 99)), "max": float(np.max(ratio)), "frames_compared": len(ratio)} if ratio else None),
           "gap_over_same_hop": ({"p99": float(np.percentile(hop_ratio, 99)), "max": float(np.max(hop_ratio)), "period": period,
                                  "frames_above_1.25x": int((np.asarray(hop_ratio) > 1.25).sum())} if hop_ratio else None),
           "frames_above_1.25x_median_gap": int((gaps > 1.25 * np.median(gaps)).sum()),
           "first_8_gaps_us": [float(x) for x in gaps[:8]]}
    if launches is not None:
        L = np.asarray(launches, float)
        out["launches_per_frame"] = {"mean": float(L[:, 0].mean()), "past_the_last_live_iteration_mean": float(L[:, 2].mean()),
                                     "note": "iteration launches enqueued per frame (icp_run_stats): k live + the one that finds out + what the "
                                             "blind prediction / run depth put behind it; rounds 1 - 3 always launched max_iterations = 40"}
    return out


def measure_tracking(icp_amd, device, hops=256):
    """Frame-to-frame tracking (README.md:4; src/ocl_icp_reg.cpp:128-172 per pair): ONE pass of `hops` frames per variant over a synthetic VGA
    sequence (five frames walked back and forth: every hop is one step of 3 degrees / (25, -10, 15) mm), frames/s and the per-frame
    distributions from that pass — cold start (every hop from the identity) and warm start (from the previous hop's T), three ways: the
    blocking icp_track_next; icp_track_submit / collect with two frames in flight from pageable host memory (the calling thread copies
    the band getLMs reads into pinned staging); the same from the engine's two pinned frame buffers (band by DMA, no host copy: two
    frames alternate)."""
    import numpy as np
    frames = [icp_amd.synth_cloud_vga(moved=f) for f in range(5)]
    order = [0, 1, 2, 3, 4, 3, 2, 1]
    seq = [frames[order[i % len(order)]] for i in range(hops + 8)]
    out = {"frames": hops, "frame": "640 x 480 float8 (9.83 MB in host memory; the 2.08 MB band getLMs reads is uploaded), 16384 landmarks, |R| = 256",
           "per_frame": "upload + getLMs + buildRBC + ICP::run to convergence (host-driven checked run: k launches, not max_iterations), result collected on the host",
           "timing": "one pass of %d frames per variant after 8 untimed frames; no pass is dropped or repeated" % hops}
    pc = time.perf_counter
    import gc
    # The timed loops run with Python's cyclic garbage collector off (as `timeit` does): a full collection of a process that has imported
    # torch takes 35 - 40 ms, and one landed in the first tracking pass of every full bench run of rounds 3 and 4 (a single 37 - 42 ms frame
    # among 256 — in the old graph path too; never in a stand-alone run of this function, which does not import torch).
    out["timing"] += "; cyclic GC of the bench process collected before and switched off during each pass"
    for name, warm in (("cold_start", False), ("warm_start", True)):
        g = icp_amd.ICP(device)
        g.init(16384, 256, ALPHA, SCALING)
        res = {}
        for f in seq[:8]:
            g.track_next(f, warm_start=warm)
        g.sync()
        ks, stamps, lat, st = [], [], [], []
        gc.collect(); gc.disable()
        t0 = pc()
        for f in seq[8:]:
            ts = pc()
            ks.append(g.track_next(f, warm_start=warm))
            te = pc()
            stamps.append(te); lat.append((te - ts) * 1e6); st.append(g.run_stats())
        el = pc() - t0
        gc.enable()
        res["blocking"] = _track_report(hops, el, np.diff(np.array([t0] + stamps)) * 1e6, lat, ks, st, period=len(order))
        res["blocking"]["host_launch_calls"] = dict(zip(("longest_us", "slower_than_10us", "calls"), g.launch_stats(reset=True)))

        def pipelined(submit_of, n):
            """Two frames in flight: submit frame i, then collect frame i - 1."""
            sub, stamps, ks, call = [], [], [], []
            gc.collect(); gc.disable()
            t0 = pc()
            for i in range(n):
                sub.append(pc())
                submit_of(i)
                call.append((pc() - sub[-1]) * 1e6)
                if i >= 1:
                    ks.append(g.track_collect()[0]); stamps.append(pc())
            ks.append(g.track_collect()[0]); stamps.append(pc())
            el = pc() - t0
            gc.enable()
            pipelined.submit_call_us = _dist(call)       # how long icp_track_submit holds the calling thread
            return el, np.diff(np.array([t0] + stamps)) * 1e6, (np.array(stamps) - np.array(sub)) * 1e6, ks

        g.track_reset()
        g.track_pipelined(seq[:8], warm_start=warm)            # (the first frame of a sequence registers against nothing: it is in the warm-up)
        g.sync()
        g.launch_stats(reset=True)
        # the warm-up's last frame is frame 7 of the sequence: the pass continues it
        tail = seq[8:]
        el, gaps, lats, ks = pipelined(lambda i: g.track_submit(tail[i], warm), hops)
        res["pipelined_pageable"] = _track_report(hops, el, gaps, lats, ks, period=len(order))
        res["pipelined_pageable"]["submit_call_us"] = pipelined.submit_call_us
        res["pipelined_pageable"]["host_launch_calls"] = dict(zip(("longest_us", "slower_than_10us", "calls"), g.launch_stats(reset=True)))
        # the same pass with the sequence's five frame buffers registered as DMA sources (icp_track_register_source: what a capture loop that
        # reuses its buffers does once): the band goes by DMA from the caller's own memory, the calling thread copies nothing
        try:
            for fr in frames:
                g.track_register(fr)
            g.track_reset()
            g.track_pipelined(seq[:8], warm_start=warm)
            g.sync()
            g.launch_stats(reset=True)
            el, gaps, lats, ks = pipelined(lambda i: g.track_submit(tail[i], warm), hops)
            res["pipelined_registered"] = _track_report(hops, el, gaps, lats, ks, period=len(order))
            res["pipelined_registered"]["submit_call_us"] = pipelined.submit_call_us
            res["pipelined_registered"]["host_launch_calls"] = dict(zip(("longest_us", "slower_than_10us", "calls"), g.launch_stats(reset=True)))
            res["pipelined_registered"]["note"] = "the caller's own frame buffers, page-locked once (icp_track_register_source): band by 2-D DMA, no host copy"
            for fr in frames:
                g.track_unregister(fr)
        except Exception as e:                       # noqa: BLE001
            res["pipelined_registered"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not warm:                                  # (two alternating frames make every warm start the inverse of what is needed: cold only)
            g.track_reset()
            g.track_staging(0)[...] = frames[1]
            g.track_staging(1)[...] = frames[2]
            for i in range(8):
                g.track_submit(i & 1, warm)
                g.track_collect()
            g.sync()
            el, gaps, lats, ks = pipelined(lambda i: g.track_submit(i & 1, warm), hops)
            res["pipelined_pinned"] = _track_report(hops, el, gaps, lats, ks, period=2)
            res["pipelined_pinned"]["submit_call_us"] = pipelined.submit_call_us
            res["pipelined_pinned"]["host_launch_calls"] = dict(zip(("longest_us", "slower_than_10us", "calls"), g.launch_stats(reset=True)))
            res["pipelined_pinned"]["note"] = ("two frames one step apart alternate in the engine's pinned frame buffers (icp_track_staging): what a capture "
                                               "loop that writes its frames there would see")
        out[name] = res
        g.close()
    return out


def measure_holes(icp_amd, device, iters):
    """`other_configs.A_holes` (+ the dense layouts beside it): the headline workload with the invalid points of a real capture in BOTH
    frames — a Kinect frame's pixels without depth are points at the origin with their colour kept (reference
    src/kinect_frame_grabber.cpp:246-262; getLMs picks them on purpose, kernels/icp_kernels.cl:49-50) — scattered and as contiguous
    regions, 10 % and 30 %; the `_rgb0` cases zero the colour too: every invalid point is then ONE point and one representative's list
    holds them all (the degenerate list).  Per case: fixed `iters`-iteration fresh passes like the headline, the longest list, the
    candidates stage 2 would have to evaluate without pruning (sum over the queries of the length of the list they land in, last
    iteration), and k of a checked run.  Plus one cold tracking pass on frames with contiguous holes."""
    import numpy as np
    from icp_amd import workloads as W
    out = {"invalid_points": "xyz = 0 in both frames (independent patterns), colour kept unless the case says rgb0; icp_synth_punch_holes",
           "timing": "per case: 25 ms of untimed passes, 3 warm-up + 20 timed fresh passes of %d iterations (HIP events on the engine's stream)" % iters}

    def one(cfg, batch, name, steps):
        side, nr = W.CONFIGS[cfg]
        m = side * side
        g = icp_amd.ICP(device)
        g.init(m, nr, ALPHA, SCALING, batch=batch)
        holes = 0.0
        for b in range(batch):
            F, M = (icp_amd.synth_pair(side, seed=W.BASE_SEED + b) if name == "clean" else W.holes_pair(icp_amd, name, side, seed=W.BASE_SEED + b))
            g.write(icp_amd.Memory.F, F, batch_index=b); g.write(icp_amd.Memory.M, M, batch_index=b)
            holes += float(np.count_nonzero((F[:, 0] == 0) & (F[:, 1] == 0) & (F[:, 2] == 0))) / m / batch
        g.buildRBC(); g.sync()
        settle(g, iters)
        for _ in range(3):
            g.run_fixed_fresh(iters)
        g.sync()
        ms, n = g.time_run_fixed_tail(iters, steps, from_identity=True)
        N, rid = g.read(icp_amd.Memory.RBC_N), g.read(icp_amd.Memory.RID)
        r = {"us_per_iteration": ms * 1e3 / (n * iters * batch), "N_max": int(N.max()), "invalid_fraction_fixed": round(holes, 4),
             "candidates_per_iteration": int(N[rid].astype(np.int64).sum())}
        if batch == 1:
            g.reset_transform(); g.buildRBC()
            r["run_k"] = int(g.run())
        g.close()
        return r

    for key, cfg, batch, names, steps in (("A_holes", "A", 1, ["clean"] + list(W.HOLES), 21),
                                          ("A_x64_holes", "A", 64, ["clean", "scattered10", "blobs10", "blobs30"], 4),
                                          ("B_holes", "B", 1, ["clean", "scattered10", "blobs10", "blobs30", "blobs30_rgb0"], 11)):
        out[key] = {}
        for name in names:
            try:
                out[key][name] = one(cfg, batch, name, steps)
            except Exception as e:                   # noqa: BLE001
                out[key][name] = {"error": "%s: %s" % (type(e).__name__, e)}
    # The reference's second example scene (data/kg_pc8d_wall, data/README.md:11-16), stand-in: a textured plane moved in its own plane
    try:
        F, M, Tt = W.wall_pair(icp_amd)
        wall = {"scene": "textured plane at 600 mm, moved %.0f degrees about its normal and %s mm in its plane; |F|=|M|=16384, |R|=256" % (W.WALL_ROT_DEG, (W.WALL_T[:2],))}
        for tag, a in (("a_2e2", ALPHA), ("a_1e-6", W.WALL_A_SMALL)):
            g = icp_amd.ICP(device)
            g.init(16384, 256, a, SCALING, max_iterations=W.WALL_MAX_ITERATIONS)
            g.write(icp_amd.Memory.F, F); g.write(icp_amd.Memory.M, M)
            g.buildRBC(); g.sync()
            settle(g, iters)
            ms, n = g.time_run_fixed_tail(iters, 21, from_identity=True)
            g.reset_transform(); g.buildRBC()
            t0 = time.perf_counter()
            k = int(g.run())
            run_ms = (time.perf_counter() - t0) * 1e3
            st = g.state()
            wall[tag] = {"us_per_iteration": ms * 1e3 / (n * iters), "N_max": int(g.read(icp_amd.Memory.RBC_N).max()),
                         "run_k": k, "run_converged": bool(st.converged), "run_ms": run_ms, "max_iterations": W.WALL_MAX_ITERATIONS,
                         "rotation_error_deg": W.rotation_error_deg(g.read(icp_amd.Memory.T), Tt), "scale": float(g.read(icp_amd.Memory.T)[7])}
            g.close()
        out["A_wall"] = wall
    except Exception as e:                           # noqa: BLE001
        out["A_wall"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # tracking: 640 x 480 frames with contiguous invalid regions (10 %, another pattern per frame), cold start, two frames in flight
    try:
        frames = [icp_amd.punch_holes(icp_amd.synth_cloud_vga(moved=f), 640, 480, icp_amd.HOLES_CONTIGUOUS, 0.1, True, seed=W.BASE_SEED + f) for f in range(5)]
        order = [0, 1, 2, 3, 4, 3, 2, 1]
        hops = 128
        seq = [frames[order[i % len(order)]] for i in range(hops + 8)]
        g = icp_amd.ICP(device)
        g.init(16384, 256, ALPHA, SCALING)
        g.track_pipelined(seq[:8], warm_start=False)
        g.sync()
        ks = []
        t0 = time.perf_counter()
        for i, f in enumerate(seq[8:]):
            g.track_submit(f, False)
            if i >= 1:
                ks.append(g.track_collect()[0])
        ks.append(g.track_collect()[0])
        el = time.perf_counter() - t0
        g.close()
        out["track_blobs10"] = {"frames": hops, "frames_per_s": hops / el, "mean_iterations": float(np.mean(ks)),
                                "form": "cold start, two frames in flight, pageable source (compare other_configs.track.cold_start.pipelined_pageable)"}
    except Exception as e:                           # noqa: BLE001
        out["track_blobs10"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


LINE_LIMIT = 4096                                # bytes of the ONE stdout line (the driver's record keeps the last 8 KB of stdout)
EXTRA_FILE = "bench_extra.json"


def _r(x, sig=6):
    """A number with `sig` significant digits (the line is read by people and a parser: 17 digits of a timing are noise)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except (TypeError, ValueError):
        return None


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full):
    """The ONE stdout line from the full measurement dictionary: the contract's fields, `roofline` and `cpu_baseline` as small objects,
    flat scalars for the other BASELINE configs — everything else stays in bench_extra.json.  Always shorter than LINE_LIMIT: optional
    scalars are dropped from the end until it is (the contract's fields never are)."""
    rl, cb, cfg = full.get("roofline") or {}, full.get("cpu_baseline"), full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(full.get("value"), 8), _r(full.get("ms_per_step"), 8)
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], "parallelism": str(cfg.get("parallelism", ""))[:80],
                      "registrations_per_gpu": cfg.get("registrations_per_gpu"), "launches_per_iteration": cfg.get("launches_per_iteration"),
                      "reduce_mode": cfg.get("reduce_mode"), "power_start": cfg.get("power_start")}
    if cfg.get("devices") is not None:
        line["config"]["devices"] = cfg["devices"]
    if full.get("roofline") is not None:
        line["roofline"] = {"bound": rl.get("bound"), "kernel": str(rl.get("kernel", ""))[:40], "achieved": _r(rl.get("achieved")),
                            "peak": rl.get("peak"), "unit": rl.get("unit"), "frac": _r(rl.get("frac")), "frac_moved": _r(rl.get("frac_moved")),
                            "traffic": _r(rl.get("traffic"), 7), "traffic_measured_by_this_run": False,
                            "algorithmic_bytes_per_launch": rl.get("algorithmic_bytes_per_launch"),
                            "avg_launch_us": _r(rl.get("avg_launch_us")), "registrations_per_launch": rl.get("registrations_per_launch"),
                            "valu_issue_frac": _r(_get(rl, "valu_beside_it", "executed", "valu_issue_frac"))}
    if cb is not None:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "threads": cb.get("threads"),
                                "host_cores": cb.get("host_cores"), "cpu_share": _r(cb.get("cpu_share")), "kind": cb.get("kind"),
                                "sample": str(cb.get("sample_short") or cb.get("sample", ""))[:160]}
    line["us_per_iteration"] = _r(full.get("us_per_iteration"))
    line["per_gpu_iterations_per_s"] = [_r(x) for x in (full.get("per_gpu_iterations_per_s") or [])]
    line["single_gpu_same_work_key"] = "config4_per_gpu_value"
    line["config4_per_gpu_value"] = _r(full.get("config4_per_gpu_value"), 8)
    line["git_head"] = full.get("git_head")
    line["extra"] = EXTRA_FILE
    oc = full.get("other_configs") or {}
    optional = []                                    # (key, value) in the order they are kept
    for key in ("A_x64", "B", "C"):
        c = oc.get(key) or {}
        optional += [("%s_us_per_iteration" % key, _r(c.get("us_per_iteration"))), ("%s_hbm_frac" % key, _r(c.get("hbm_frac"))),
                     ("%s_valu_issue_frac" % key, _r(c.get("valu_issue_frac"))), ("%s_k_search_us" % key, _r(c.get("k_search_avg_launch_us"))),
                     ("%s_build_rbc_ms" % key, _r(c.get("build_rbc_ms")))]
        if "error" in c:
            optional.append(("%s_error" % key, str(c["error"])[:80]))
    optional += [("reference_order_us_per_iteration", _r(full.get("reference_order_us_per_iteration"))),
                 ("reference_order_squared_us_per_iteration", _r(full.get("reference_order_squared_us_per_iteration"))),
                 ("value_right_after_start", _r(_get(full, "value_right_after_start", "iterations_per_s"))),
                 ("build_rbc_ms", _r(_get(full, "registration_latency", "build_rbc_ms"))),
                 ("identical_ids_frac_vs_reference_order", _r(_get(full, "mode_note", "identical_ids_frac"))),
                 ("dt_over_t_vs_reference_order", _r(_get(full, "mode_note", "vs_float64", "between_the_modes_fp32", "dt_over_t"))),
                 ("dt_over_t_vs_float64", _r(_get(full, "mode_note", "vs_float64", "benchmarked", "dt_over_t")))]
    for name, path in (("track_cold_frames_per_s", ("track", "cold_start", "pipelined_pageable", "frames_per_s")),
                       ("track_warm_frames_per_s", ("track", "warm_start", "pipelined_registered", "frames_per_s")),
                       ("track_warm_gap_p99_over_same_hop", ("track", "warm_start", "pipelined_registered", "gap_over_same_hop", "p99")),
                       ("A_scattered10_us_per_iteration", ("holes", "A_holes", "scattered10", "us_per_iteration")),
                       ("A_blobs30_us_per_iteration", ("holes", "A_holes", "blobs30", "us_per_iteration")),
                       ("A_blobs10_rgb0_us_per_iteration", ("holes", "A_holes", "blobs10_rgb0", "us_per_iteration")),
                       ("A_blobs30_rgb0_us_per_iteration", ("holes", "A_holes", "blobs30_rgb0", "us_per_iteration")),
                       ("A_wall_us_per_iteration", ("holes", "A_wall", "a_2e2", "us_per_iteration")),
                       ("B_blobs30_us_per_iteration", ("holes", "B_holes", "blobs30", "us_per_iteration")),
                       ("B_blobs30_rgb0_us_per_iteration", ("holes", "B_holes", "blobs30_rgb0", "us_per_iteration")),
                       ("A_x64_blobs30_us_per_iteration", ("holes", "A_x64_holes", "blobs30", "us_per_iteration"))):
        optional.append((name, _r(_get(oc, *path))))
    for k, v in optional:
        if v is not None:
            line[k] = v
    text = json.dumps(line, separators=(", ", ": "))
    keys = [k for k, v in optional if v is not None]
    while len(text.encode()) >= LINE_LIMIT and keys:
        line.pop(keys.pop())
        text = json.dumps(line, separators=(", ", ": "))
    if len(text.encode()) >= LINE_LIMIT:            # (cannot happen with the bounded strings above: fail loudly rather than print a line nobody parses)
        raise SystemExit("bench.py: the result line is %d bytes, limit %d" % (len(text.encode()), LINE_LIMIT))
    return text


def emit(full, out=None, err=None, extra_dirs=None):
    """Everything measured -> bench_extra.json (beside bench.py and, on a GPU box, under gpurun_out/), a one-line pointer on stderr;
    the compact line -> the LAST line of stdout."""
    out, err = out or sys.stdout, err or sys.stderr
    text = compact_line(full)
    blob = json.dumps(full)
    for d in (extra_dirs if extra_dirs is not None else (ROOT, os.path.join(ROOT, "gpurun_out"))):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, EXTRA_FILE), "w") as f:
                f.write(blob + "\n")
        except OSError:
            pass
    # stderr gets a pointer, not the record: whoever keeps only the tail of the two streams together must still find the line in it
    # (ICP_BENCH_STDERR_RECORD=1: the whole record on stderr as well)
    if os.environ.get("ICP_BENCH_STDERR_RECORD") == "1":
        err.write("bench.py: full measurement record (%d bytes, also in %s):\n%s\n" % (len(blob), EXTRA_FILE, blob))
    else:
        err.write("bench.py: full measurement record (%d bytes) in %s; the result line (%d bytes) is the last line of stdout\n" % (len(blob), EXTRA_FILE, len(text)))
    err.flush()
    out.write(text + "\n")
    out.flush()
    return text


def run_inprocess(icp_amd, args, n, batch, iters, steps, warmup):
    """--gpus N from a plain `python bench.py`: devices 0..N-1 (or ICP_BENCH_DEVICES, a comma list: self-test on a 1-GPU box)
    driven through icp_batch_*: registration i on slot i mod N, one host thread + stream per slot, no collective."""
    devs = os.environ.get("ICP_BENCH_DEVICES")
    devices = [int(x) for x in devs.split(",")] if devs else list(range(n))
    if len(devices) != n:
        raise SystemExit("bench.py: --gpus %d but ICP_BENCH_DEVICES names %d devices" % (n, len(devices)))
    have = icp_amd.device_count()
    if have <= max(devices):
        raise SystemExit("bench.py: --gpus %d needs device ordinal %d, but only %d device(s) are visible" % (n, max(devices), have))
    from icp_amd import workloads as W
    side, nr = W.CONFIGS[args.config]
    m = side * side
    B = icp_amd.ICPBatch(devices)
    B.init(n * batch, m, nr, ALPHA, SCALING)
    B.set_modes(icp_amd.ReduceMode.FUSED if args.reduce_mode == "fused" else icp_amd.ReduceMode.REFERENCE_ORDER,
                icp_amd.PowerMode.SQUARED if args.power_mode == "squared" else icp_amd.PowerMode.LITERAL)
    for i in range(n * batch):
        F, M = pair_of(icp_amd, args.config, batch, i)
        B.write(i, icp_amd.Memory.F, F)
        B.write(i, icp_amd.Memory.M, M)
    B.buildRBC()
    seconds, slot_ms = B.time_run_fixed_slots(iters, steps, warmup + 5)      # (+ 5 untimed passes per slot: the set-up's settle (), see there)
    B.close()
    per_gpu = [float(steps * iters * batch / (float(ms) * 1e-3)) for ms in slot_ms]
    return seconds, steps * iters * batch * n, per_gpu, devices


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the B / C / tracking / mode measurements of the N = 1 line")
    ap.add_argument("--power-mode", choices=["literal", "squared"], default="squared")
    ap.add_argument("--reduce-mode", choices=["reference", "fused"], default="fused")
    ap.add_argument("--config", choices=["A", "B", "C"], default="A",
                    help="workload of the line: A = BASELINE configs[1] (default, the metric), B = configs[2], C = configs[4]")
    ap.add_argument("--batch", type=int, default=0,
                    help="independent registrations per GPU sharing each launch; default: 1 on one GPU (the headline metric), "
                         "64 per GPU on several (BASELINE config 4)")
    args = ap.parse_args()

    launch, world = resolve_launch(args.gpus, os.environ)
    rank = int(os.environ.get("RANK", "0")) if launch == "ranks" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launch == "ranks" else 0
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
    if launch == "ranks":
        if torch is None:
            raise SystemExit("bench.py: torch.distributed is required for a multi-rank launch")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo by default: the engine's contract is "no RCCL" (replicas only) — torch.distributed carries a barrier and
        # three 8-byte reductions of host numbers, nothing on the data path.  ICP_BENCH_BACKEND=nccl (= RCCL) is accepted; whichever
        # backend is asked for, a failed init falls back to the other one here, before any engine call.
        backend = os.environ.get("ICP_BENCH_BACKEND", "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("ICP_BENCH_DEVICE", local_rank)))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:                       # noqa: BLE001 — whatever the backend raises
            other = "gloo" if backend != "gloo" else "nccl"
            sys.stderr.write("bench.py: init_process_group (%s) failed (%s): trying %s\n" % (backend, e, other))
            if dist.is_initialized():
                dist.destroy_process_group()
            if other == "nccl" and not torch.cuda.is_available():
                raise
            backend = other
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))

    import icp_amd
    device = int(os.environ.get("ICP_BENCH_DEVICE", local_rank))   # override: self-test of the N>1 path on a 1-GPU box
    host_cpus = None
    if launch == "ranks" and os.environ.get("ICP_AMD_SLOT_NUMA", "1") != "0":
        # a rank drives one GPU from this thread: keep it on that GPU's NUMA node (what icp_batch_create does for its slot threads; a two-socket
        # 8-GPU node has four GPUs per socket).  Silently skipped where sysfs says nothing or none of those CPUs is ours.
        try:
            cl = icp_amd.numa_cpulist(icp_amd.device_pci_bus_id(device))
            want = set()
            for part in cl.split(","):
                if part:
                    a, _, b = part.partition("-")
                    want.update(range(int(a), int(b or a) + 1))
            mine = want & set(os.sched_getaffinity(0))
            if mine:
                os.sched_setaffinity(0, mine)
                host_cpus = cl
        except Exception:                            # noqa: BLE001 — placement is an optimisation, never a reason to fail
            host_cpus = None
    fused = args.reduce_mode == "fused"

    per_gpu = None
    devices_used = None
    setup_passes = 0
    if launch == "inprocess":
        total_t, total_iters, per_gpu, devices_used = run_inprocess(icp_amd, args, world, batch, iters, steps, warmup)
        device = devices_used[0]
        # the dominant kernel's launch time, live, on the first device: a handle of its own with the same per-GPU work
        g, m, nr, (F, M) = setup(icp_amd, device, args.config, batch, 0, args.power_mode, args.reduce_mode)
        rsteps = max(2, min(steps, 10))
        g.run_fixed_fresh(iters)
        g.sync()
        ev_ms, ev_steps = g.time_run_fixed_tail(iters, rsteps, from_identity=True)
    else:
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

        # what a freshly started process sees (reported beside the metric, never as it): the same W warm-up + K timed steps BEFORE the
        # device has been kept busy for SETUP_MS — an MI355X that was idle runs its first ~10 ms of work below its steady clocks
        unsettled = None
        if dist is None:
            for _ in range(warmup):
                g.run_fixed_fresh(iters)
            g.sync()
            t0u = time.perf_counter()
            g.time_run_fixed_tail(iters, steps, from_identity=True)
            g.sync()
            unsettled = steps * iters * batch / (time.perf_counter() - t0u)
        setup_passes = settle(g, iters)
        for _ in range(warmup):
            g.run_fixed_fresh(iters)                     # a fresh registration: from the identity transform, one graph
        barrier()
        t0 = time.perf_counter()
        # the K steps; the engine brackets steps 2 .. K with hipEvents on its own stream (roofline duration: a marker recorded on an idle
        # stream would hold the first graph back by 0.1 - 0.25 ms, icp_time_run_fixed_tail); the wall clock covers all K
        ev_ms, ev_steps = g.time_run_fixed_tail(iters, steps, from_identity=True)
        barrier()
        elapsed = time.perf_counter() - t0
        total_t, total_iters = aggregate(dist, elapsed, steps * iters * batch)
        per_gpu = gather_per_rank(dist, steps * iters * batch / elapsed)

    # beside the metric (never part of `value`): latency of one whole registration = RBC construction + the iterations,
    # inputs resident, and the same with the two clouds uploaded from host memory first
    e2e = None
    if rank == 0 and batch == 1 and launch == "single":
        reps = 20 if m <= 65536 else 3
        g.buildRBC(); g.sync()
        t1 = time.perf_counter()
        for _ in range(reps):
            g.buildRBC(); g.run_fixed_fresh(iters)
        g.sync()
        resident_ms = (time.perf_counter() - t1) / reps * 1e3
        upload_ms = None
        for _ in range(2):                           # (the first round of interleaved copies and graphs runs at half speed: warm-up)
            g.sync()
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
    # `iters` k_search launches (+ one end kernel per graph); otherwise a graph holding only that kernel.
    tkey = "k_search_hbm_bytes_per_launch" if (args.config == "A" and batch == 1) else \
           "k_search_hbm_bytes_per_launch_%s" % (args.config if batch == 1 else "%s_x%d" % (args.config, batch))
    roofline = roofline_of(g, m, nr, batch, iters, ev_steps, ev_ms, fused, tkey) if rank == 0 else None
    launches = g.launches_per_iteration()
    headline = launch == "single" and args.config == "A" and batch == 1
    modes = None
    if rank == 0 and headline and fused and args.power_mode == "squared" and not args.no_other_configs:
        try:
            modes = measure_modes(icp_amd, device, g, args.power_mode, args.reduce_mode)
        except Exception as e:                       # noqa: BLE001 — beside the metric: reported, never fatal
            modes = {"mode_note": {"error": "%s: %s" % (type(e).__name__, e)}}
    g.close()

    if rank == 0:
        cfg_text = {"A": "configs[1]: synthetic kg-like pair, |F|=|M|=16384, |R|=256",
                    "B": "configs[2]: synthetic VGA RGB-D cloud subsampled to |F|=|M|=65536, |R|=1024",
                    "C": "configs[4]: single registration |F|=|M|=2^20, |R|=4096"}[args.config]
        metric = "ICP iterations/sec at |F|=|M|=%d, |R|=%d" % (m, nr)
        if args.config == "A" and batch > 1:
            cfg_text = "configs[3]: independent frame pairs of |F|=|M|=16384, |R|=256, %d per GPU sharing each launch, no RCCL" % batch
        if batch > 1:
            metric += ", %d independent registrations per GPU per launch" % batch
        line = {
            "metric": metric,
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
                       "parallelism": "single" if world == 1 else ("replicas: one rank per GPU (torch.distributed.run; %s for the barrier and the reductions of the report, no collective on the data path)" % dist.get_backend() if launch == "ranks" else
                                                                   "replicas: icp_batch_* in-process, one host thread + stream per GPU"),
                       "registrations_per_gpu": batch, "power_start": args.power_mode,
                       "setup": ("RBC built, the step's graph instantiated and run %d times (%.0f ms, untimed: clocks of a device that was idle) before the "
                                 "%d warm-up steps" % (setup_passes, SETUP_MS, warmup)) if launch != "inprocess" else
                                ("RBC built, the step's graph instantiated and run 5 times per device (untimed) before the %d warm-up steps" % warmup),
                       "reduce_mode": args.reduce_mode, "launches_per_iteration": launches},
            "roofline": roofline,
        }
        if launch == "single":
            line["value_right_after_start"] = {"iterations_per_s": unsettled,
                                               "note": "the same %d warm-up + %d timed steps measured BEFORE the %.0f ms of untimed set-up work (a device that has just been "
                                                       "idle, below its steady clocks): what the first registrations of a freshly started process run at; `value` is "
                                                       "the steady state" % (warmup, steps, SETUP_MS)}
        line["per_gpu_iterations_per_s"] = per_gpu
        if devices_used is not None:
            line["config"]["devices"] = devices_used
        if host_cpus:
            line["config"]["rank0_host_cpus"] = host_cpus
        # the same-work field of every line: one GPU at config 4's per-GPU share (64 registrations of A per launch)
        line["single_gpu_same_work_key"] = "config4_per_gpu_value"
        line["config4_per_gpu_value"] = None
        if args.config == "A" and batch == 64 and world >= 1:
            line["config4_per_gpu_value"] = total_iters / total_t / world
        if world > 1:
            line["config"]["scaling_reference"] = ("per-GPU work is %d registrations per launch; `config4_per_gpu_value` of this line = value / n_gpus, "
                                                   "of the N = 1 line = the same 64-registration work measured on one GPU: compare those, not the "
                                                   "N = 1 batch-1 headline `value`" % batch)
        if e2e is not None:
            line["registration_latency"] = e2e
        if headline:
            # config 4's per-GPU share on this GPU (always: the field every line carries), then the other BASELINE configs, same step
            # definition, fewer steps (C: 3 steps of 10 iterations), tracking, and what the benchmarked mode is against the other one
            # (everything beside the metric is guarded: a failure there is reported in its place and never costs the line its `value`)
            def guarded(fn, *a, **kw):
                try:
                    return fn(*a, **kw)
                except Exception as e:               # noqa: BLE001
                    return {"error": "%s: %s" % (type(e).__name__, e)}

            oc = {"A_x64": guarded(measure_config, icp_amd, device, "A", 64, 20, 3, ITERS_PER_STEP, args.power_mode, args.reduce_mode, warm_seed=not args.no_other_configs)}
            line["config4_per_gpu_value"] = oc["A_x64"].get("iterations_per_s")
            if not args.no_other_configs:
                for key, cfg, b, st, wu, it in (("B", "B", 1, 40, 5, ITERS_PER_STEP), ("C", "C", 1, 10, 2, 10)):
                    oc[key] = guarded(measure_config, icp_amd, device, cfg, b, st, wu, it, args.power_mode, args.reduce_mode)
                oc["track"] = guarded(measure_tracking, icp_amd, device)
                oc["holes"] = guarded(measure_holes, icp_amd, device, ITERS_PER_STEP)
            line["other_configs"] = oc
            if modes is not None:
                line.update(modes)
        if world == 1 and not args.no_cpu_baseline and m <= 65536:
            line["cpu_baseline"] = cpu_baseline(F, M, m, nr, fused)
        line["git_head"] = git_head()
        emit(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
