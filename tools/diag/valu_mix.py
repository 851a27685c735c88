"""Diagnostic (needs hipcc only): the VALU instruction mix of a kernel's loops, priced with the measured issue costs of
tests/cpp/valu_issue_probe.hip (profiles/valu_issue.json: SIMD cycles per wave64 instruction by class).  Static: every instruction
that sits inside a loop (a backward branch) counts once — the dynamic count is dominated by the loops, and their classes' shares are
what the average cost depends on.  SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU (profiles/traffic.json) is printed beside it: 4.00 for every kernel — that counter holds one
quad-cycle per instruction whatever its class (profiles/r06_valu_classes.txt: 1.0 for every probe class but the transcendentals' 2.0), so
round 5's "the counters side with 4 cycles" was circular.
usage: python tools/diag/valu_mix.py [SOURCE NAME-SUBSTRING [WAVES-PER-SIMD]]      (default: the four bench kernels)"""
import collections, json, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import FLAGS, HIPCC, ROOT

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32"}          # measured: v_fma / add / mul / sub / fmac f32, add_u32, and_b32, mov_b32; the rest of the row by kinship


def costs(path=os.path.join(ROOT, "profiles", "valu_issue.json")):
    return json.load(open(path))


def price(mn, ops, C, w):
    """(cycles, class) of one instruction; w = waves per SIMD column of the probe ("1", "2", "4", "8")."""
    g = lambda k: C[k][w]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if mn.startswith("v_mfma") or mn.startswith("v_accvgpr"):
        return 0.0, "mfma / accvgpr (own pipe)"
    if "dpp" in mn or "dpp" in ops or "quad_perm" in ops or "row_" in ops:
        return g("v_mov_b32_dpp"), "dpp"
    if base.startswith("v_pk_"):
        return g("v_pk_fma_f32"), "packed"
    if base.endswith("_f64") or base in ("v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_f64_u32", "v_cvt_f64_i32", "v_ldexp_f64"):
        return g("v_fma_f64"), "f64"
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return g("v_rcp_f32"), "transcendental"
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
        return g("v_readlane_b32"), "lane <-> scalar"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        to_vcc = ops.split(",")[0].strip() == "vcc" or mn.endswith("_e32")
        return (g("v_cmp_lt_f32") if to_vcc else g("v_cmp_lt_f32 (e64, SGPR pair)")), "compare"
    if base == "v_cndmask_b32":
        if ops.rstrip().endswith("vcc"):
            # behind the compare that wrote vcc: the measured pair minus the compare
            return 2.0 * g("v_cmp_lt_f32+v_cndmask_b32 (pair)") - g("v_cmp_lt_f32"), "select (vcc)"
        return g("v_cndmask_b32 (e64, SGPR mask)"), "select (SGPR mask)"
    if base in C:
        return g(base), "fast f32 / int" if base in FAST else "other full-rate-4"
    if base in FAST:
        return g("v_add_f32"), "fast f32 / int"
    return g("v_and_or_b32"), "other full-rate-4"


def loops_of(source, pat):
    """[(kernel name, [(mnemonic, operands, in_loop)])] of the kernels whose mangled name contains `pat`."""
    p = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-o", "-", os.path.join(ROOT, source)], capture_output=True, text=True, cwd=ROOT)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-3000:])
    out, cur, name = [], None, None
    for line in p.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and ".type" not in line:
            name, cur = m.group(1), []
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end") and cur is not None:
            if pat in name:
                out.append((name, cur))
            cur = None
        elif cur is not None and t and not t.startswith(";"):
            if re.match(r"^\.LBB\d+_\d+:", t):
                cur.append(("label", t.split(":")[0], None))
            elif not t.startswith("."):
                t = t.split(";")[0].strip()
                mn, _, ops = t.partition(" ")
                cur.append((mn, ops.strip(), None))
    res = []
    for name, ins in out:
        pos = {x[1]: i for i, x in enumerate(ins) if x[0] == "label"}
        depth = [0] * len(ins)
        for i, (mn, ops, _) in enumerate(ins):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                tgt = ops.split()[-1]
                if tgt in pos and pos[tgt] < i:
                    for j in range(pos[tgt], i + 1):
                        depth[j] += 1
        res.append((name, [(mn, ops, depth[i] > 0) for i, (mn, ops, _) in enumerate(ins) if mn != "label"]))
    return res


def mix(source, pat, w="8", verbose=True):
    C = costs()
    rows = []
    for name, ins in loops_of(source, pat):
        for scope in ("loops", "all"):
            cls, tot, cyc, top = collections.Counter(), 0, 0.0, collections.Counter()
            clc = collections.Counter()
            for mn, ops, inloop in ins:
                if not mn.startswith("v_") or (scope == "loops" and not inloop):
                    continue
                c, k = price(mn, ops, C, w)
                if c == 0.0:
                    continue
                tot += 1; cyc += c; cls[k] += 1; clc[k] += c; top[re.sub(r"_(e32|e64)$", "", mn)] += 1
            if not tot:
                continue
            rows.append((name, scope, tot, cyc / tot, cls, clc, top))
            if verbose:
                print("%s  [%s]  %d VALU instructions, %.3f cycles per instruction at %s waves / SIMD" % (name, scope, tot, cyc / tot, w))
                for k, n in cls.most_common():
                    print("      %-22s %5d  (%4.1f %% of the instructions, %4.1f %% of the cycles)" % (k, n, 100.0 * n / tot, 100.0 * clc[k] / cyc))
                if scope == "loops":
                    print("      top: " + ", ".join("%s %d" % kv for kv in top.most_common(12)))
    return rows


BENCH_KERNELS = {   # config -> (source, name substring, waves per SIMD of the variant: 16 waves / block, 1 block per CU = 4; 8 waves / block x 4 blocks = 8)
    "A": ("icp_amd/csrc/icp_kernels.hip", "k_searchILb1ELb1ELi2ELi16ELb0ELi1ELi1024ELb0ELb0ELb0E", "4"),
    "B": ("icp_amd/csrc/icp_search_dense.hip", "k_searchILb1ELb0ELi4ELi8ELb0ELi1ELi256ELb0ELb0ELb0E", "8"),
    "C": ("icp_amd/csrc/icp_search_dense.hip", "k_searchILb1ELb0ELi4ELi8ELb0ELi1ELi256ELb0ELb1ELb0E", "8"),
    "Ax64": ("icp_amd/csrc/icp_search_dense.hip", "k_searchILb1ELb0ELi4ELi8ELb0ELi1ELi256ELb1ELb0ELb0E", "8"),
}

if __name__ == "__main__":
    if len(sys.argv) > 2 and "--write" not in sys.argv:
        mix(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "8")
        sys.exit(0)
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    summary = {}
    for cfg, (src, pat, w) in BENCH_KERNELS.items():
        rows = mix(src, pat, w)
        loops = [r for r in rows if r[1] == "loops"]
        det = traffic.get("detail_" + cfg, {})
        measured = det["SQ_ACTIVE_INST_VALU_quad"] * 4.0 / det["SQ_INSTS_VALU"] if det.get("SQ_ACTIVE_INST_VALU_quad") else None
        if loops:
            summary[cfg] = {"kernel": loops[0][0], "waves_per_simd": int(w), "cycles_per_valu_instruction_from_the_mix": round(loops[0][3], 3),
                            "sq_active_inst_valu_x4_per_instruction": round(measured, 3) if measured else None}
            print("==> %s: mix %.3f cycles per VALU instruction; SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU: %s (that counter holds one quad-cycle per instruction whatever its class — the probe's kernels all read 1.0 or 2.0 —: it is a count, not a cost)\n" % (cfg, loops[0][3], "%.3f" % measured if measured else "n/a"))
    print(json.dumps(summary, indent=1))
    if "--write" in sys.argv:                        # profiles/valu_issue.json: the probe's constants + what they make of the bench kernels
        path = os.path.join(ROOT, "profiles", "valu_issue.json")
        J = json.load(open(path))
        J["_kernel_mix"] = dict(summary, _how="tools/diag/valu_mix.py: the VALU instructions inside the loops of the kernel's ISA, each priced with the probe's "
                                              "constant of its class at the variant's waves per SIMD; bench.py prices SQ_INSTS_VALU with the average")
        json.dump(J, open(path, "w"), indent=1)
