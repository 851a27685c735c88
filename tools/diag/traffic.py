"""Diagnostic (not a test): writes profiles/traffic.json and copies the round's profile summaries into profiles/ from the
output of tools/diag/profiles.sh (gpurun_out/<TAG>_*).  usage: python tools/diag/traffic.py r03 COMMIT   (COMMIT = the engine commit the passes were measured at)"""
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = json.load(open(os.path.join(ROOT, "gpurun_out", "%s_profile_summary.json" % tag)))
ALGO = {"A": 72 * 16384 + 32 * 256 + 64, "B": 72 * 65536 + 32 * 1024 + 64, "C": 72 * (1 << 20) + 32 * 4096 + 64}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_* / TCC_* in separate passes (each with --kernel-trace only) around "
               "`python3 bench.py --no-cpu-baseline --no-other-configs [--config B|C | --batch 64]` (tools/diag/profiles.sh); per-dispatch means of "
               "the dominant kernel.  Units: FETCH_SIZE / WRITE_SIZE are KB (MI355X_MICROARCH.md, HBM section): hbm_bytes = (2 x FETCH_SIZE + "
               "WRITE_SIZE) x 1024 — the guide's gfx950 correction (FETCH_SIZE counts half of the bytes of a wide coalesced stream) applied; it is "
               "calibrated for 16 B / lane coalesced streams only and k_search mixes those with 16-byte gathers, so the corrected figure is an "
               "upper bound (the uncorrected one is kept beside it).  The working sets of A, B and A x 64 are Infinity-Cache resident: these "
               "fabric-side counters include Infinity Cache hits.",
       "_source": "profiles/%s_profile_summary.json" % tag,
       "_commit": (sys.argv[2] if len(sys.argv) > 2 else None)}
for cfg, key, kname, batch in (("A", "k_search_hbm_bytes_per_launch", "k_search<true, true", 1), ("B", "k_search_hbm_bytes_per_launch_B", "k_search<true, false, 4, 8, false", 1),
                               ("C", "k_search_hbm_bytes_per_launch_C", "k_search<true, false, 4, 8, false", 1),
                               ("Ax64", "k_search_hbm_bytes_per_launch_A_x64", "k_search<true, false, 4, 8, false", 64)):
    if cfg not in S:
        continue
    c = next(v for k, v in S[cfg]["counters_per_dispatch"].items() if k.startswith(kname))
    kn = next(k for k in S[cfg]["counters_per_dispatch"] if k.startswith(kname))
    st = S[cfg].get("kernel_stats", {}).get(kn, {})
    algo = ALGO["A" if cfg == "Ax64" else cfg] * batch
    out[key] = c["hbm_bytes_per_launch"]
    out["detail_" + cfg] = {
        "kernel": kn, "dispatches_counted": c["dispatches"], "avg_dispatch_us_kernel_trace": st.get("avg_us"),
        "FETCH_SIZE_KB_raw_mean": c["FETCH_SIZE"], "WRITE_SIZE_KB_mean": c["WRITE_SIZE"],
        "hbm_bytes_per_launch": c["hbm_bytes_per_launch"], "hbm_bytes_per_launch_uncorrected": c["hbm_bytes_per_launch_uncorrected"],
        "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": c["hbm_bytes_per_launch"] / algo,
        "l2_hit_rate": c.get("l2_hit_rate"), "wait_fraction_of_wave_cycles": c.get("wait_fraction_of_wave_cycles"),
        "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_VMEM_RD": c.get("SQ_INSTS_VMEM_RD"), "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
        "SQ_WAVE_CYCLES_quad": c.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY_quad": c.get("SQ_WAIT_ANY"), "SQ_ACTIVE_INST_VALU_quad": c.get("SQ_ACTIVE_INST_VALU"),
        "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES"), "SQ_WAIT_INST_ANY_quad": c.get("SQ_WAIT_INST_ANY")}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(ROOT, "gpurun_out", "%s_profile_summary.json" % tag), os.path.join(ROOT, "profiles", "%s_profile_summary.json" % tag))
for cfg in ("A", "B", "C", "Ax64", "REF"):
    src = os.path.join(ROOT, "gpurun_out", "%s_%s_kernel_stats.csv" % (tag, cfg))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, "reference_order" if cfg == "REF" else cfg)))
for k, v in out.items():
    if k.startswith("detail_"):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
