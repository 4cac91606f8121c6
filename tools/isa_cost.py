"""Issue-cycle estimate of a stretch of gfx950 ISA (hipcc -S output) from the per-class costs measured by
tools/ubench/valu_rate.hip on MI355X (cycles per wave64 instruction per SIMD with >= 4 waves resident):
  plain fp32 add/mul/fma/fmac/mov/sub ........ 2      packed fp32 (v_pk_*), DPP, v_cmp*, v_cndmask, v_min/max/med3, cvt,
  integer/logic VALU, v_readlane, v_mad_u64 ... 4      transcendental (v_exp/log/rcp/rsq/sqrt), v_permlane*_swap ........ 8
usage: python tools/isa_cost.py file.s kernel_substring [first_label last_label]"""
import re, sys, collections

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_mac_f32", "v_fmaak_f32", "v_fmamk_f32",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
SLOW8 = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_permlane16_swap", "v_permlane32_swap", "v_rcp_iflag")


def cost(op, line):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith(SLOW8):
        return 8, "trans/swap"
    if "dpp" in op or " quad_perm" in line or " row_" in line:
        return 4, "dpp"
    if base in FAST:
        return 2, "fast"
    if base.startswith("v_pk_"):
        return 4, "packed"
    if base.startswith("v_cmp") or base.startswith("v_cndmask"):
        return 4, "cmp/sel"
    return 4, "other-valu"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lo = sys.argv[3] if len(sys.argv) > 3 else None
    hi = sys.argv[4] if len(sys.argv) > 4 else None
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().endswith(":") or (l.startswith("_Z") and kern in l and ": " in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    if lo:
        a = next(i for i, l in enumerate(body) if l.startswith(lo + ":"))
        b = next(i for i, l in enumerate(body) if l.startswith(hi + ":")) if hi else len(body)
        body = body[a:b]
    tot = collections.Counter()
    n = collections.Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c, k = cost(op, t)
            tot[k] += c
            n[k] += 1
        elif op.startswith("s_"):
            n["salu"] += 1
        elif op.startswith("ds_"):
            n["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            n["vmem"] += 1
    print("instructions:", dict(n))
    print("VALU issue cycles by class:", dict(tot), "total", sum(tot.values()))


main()
