"""Diagnostic: condenses the PMC passes of tools/diag/valu_classes.sh.  (1) Per probe kernel (one instruction class each): counter value per
executed wave-instruction — which SQ_INSTS_VALU_* counter a class lands in and how it is weighted.  (2) Per bench config: the dominant
kernel's counters per dispatch.  Writes gpurun_out/<TAG>_valu_classes.json."""
import collections, csv, glob, json, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
NAMES = ["v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_f32", "v_fmac_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_min_f32_dpp", "v_mov_b32_dpp", "v_cndmask_b32",
         "v_cmp_lt_f32", "v_cmp_lt_f32+v_cndmask_b32 (pair)", "v_fma_f64", "v_add_f64", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_mov_b32", "v_mad_u32_u24", "v_sub_co_u32",
         "v_rcp_f32", "v_sqrt_f32", "v_readlane_b32", "v_cndmask_b32 (e64, SGPR mask)", "v_min_f32", "v_min3_f32", "v_sub_f32", "v_min_u32", "v_cmp_lt_f32 (e64, SGPR pair)",
         "v_bfe_u32", "v_and_or_b32", "v_mul_lo_u32", "search mix"]


def rows(pattern):
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            yield r


out = {"probe": {}, "bench": {}}
# ---- the probe: kernels k_probe<OP>; every (OP, waves-per-SIMD) is launched twice; per dispatch: waves x trips x per_trip wave-instructions
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows("gpurun_out/%s_vc_probe_*/*/*counter_collection.csv" % tag):
    m = re.search(r"k_probe<(\d+)>", r["Kernel_Name"])
    if not m:
        continue
    op = int(m.group(1))
    grid, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
    waves = grid // 64
    per_trip, trips = (160, 400) if op == len(NAMES) - 1 else (32, 2000)
    acc[op][r["Counter_Name"]].append(float(r["Counter_Value"]) / (waves * trips * per_trip))
for op in sorted(acc):
    out["probe"][NAMES[op] if op < len(NAMES) else str(op)] = {k: round(sum(v) / len(v), 4) for k, v in sorted(acc[op].items())}
# ---- the bench's dominant kernel
for cfg in ("A", "B", "C", "Ax64"):
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows("gpurun_out/%s_vc_%s_*/*/*counter_collection.csv" % (tag, cfg)):
        kn = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "k_search" in kn and (cfg != "A" or kn.startswith("k_search<true, true")):      # (A: the chained latency variant; its run also times A x 64)
            ctr[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    best = max(ctr.items(), key=lambda kv: sum(kv[1].get("SQ_INSTS_VALU", [0])), default=None)
    if best:
        out["bench"][cfg] = dict({"kernel": best[0], "dispatches": max(len(v) for v in best[1].values())}, **{k: sum(v) / len(v) for k, v in sorted(best[1].items())})
json.dump(out, open("gpurun_out/%s_valu_classes.json" % tag, "w"), indent=1)
print("counter value per executed wave-instruction, by probe class:")
keys = sorted({k for v in out["probe"].values() for k in v})
for name, v in out["probe"].items():
    print("  %-36s %s" % (name[:36], "  ".join("%s %.3g" % (k.replace("SQ_INSTS_VALU_", "").replace("SQ_", ""), x) for k, x in v.items() if x > 0.004)))
for cfg, v in out["bench"].items():
    print("==", cfg, v["kernel"][:70])
    print("   ", {k: round(x, 1) for k, x in v.items() if isinstance(x, float)})
