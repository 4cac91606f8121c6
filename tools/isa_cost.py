"""Issue-cycle estimate of a stretch of gfx950 ISA (hipcc -S output) from the per-class costs measured by
tools/ubench/valu_rate.hip on MI355X (cycles per wave64 instruction per SIMD with >= 4 waves resident):
  plain fp32 add/mul/fma/fmac/mov/sub ........ 2      packed fp32 (v_pk_*), DPP, v_cmp*, v_cndmask, v_min/max/med3, cvt,
  integer/logic VALU, v_readlane, v_mad_u64 ... 4      transcendental (v_exp/log/rcp/rsq/sqrt), v_permlane*_swap ........ 8
usage: python tools/isa_cost.py file.s kernel_substring [first_label last_label]
       python tools/isa_cost.py --bwd-model file.s      -> JSON: per-part VALU instructions / issue cycles of k_composite_bwd<1,false>
       (the table bench.py's roofline.compute multiplies the kernel's work counters with; __graft_entry__.build() regenerates it
       from the compiler's output of the shipped source into instantsplat_amd/lib/bwd_issue_model.json)"""
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


def bwd_model(path):
    """Per-part costs of the backward composite kernel's replay loop, from its basic blocks.  A block is recognised by what it
    contains: v_exp_f32 = a quadrant body up to the `any lane valid` test ("quad"), v_rcp_f32 inside the loop = the rest of the
    body ("quad_valid"), the LDS write .. atomic group = the reduction ("reduce"), blocks made of nothing but v_mov (the
    moments' initialisation the compiler peeled out of the first quadrant body: executed when that body does not run) =
    "init", everything else inside the loop = "step"; everything outside the loop = "wave" (prologue / epilogue, once per
    unit).  The loop is unrolled by two (ping-pong record registers): every part is averaged over its copies."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_composite_bwdILi1ELb0ELb0" in l and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], []
    for l in lines[start:end + 1]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur); cur = []
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
        if t.split()[0].startswith("s_cbranch") or t.split()[0] == "s_branch":
            blocks.append(cur); cur = []
    blocks.append(cur)
    has = lambda b, op: any(t.split()[0].startswith(op) for t in b)
    valu = lambda b: [t for t in b if t.startswith("v_")]
    first = next(i for i in range(len(blocks) - 1) if has(blocks[i], "s_flbit") and has(blocks[i + 1], "v_exp_f32"))
    last = max(i for i, b in enumerate(blocks) if has(b, "global_atomic"))
    parts = {k: [0, 0] for k in ("step", "quad", "quad_valid", "reduce", "init", "wave")}
    copies = sum(1 for b in blocks[first:last + 1] if has(b, "ds_write"))
    in_reduce = False
    for i, b in enumerate(blocks):
        v = valu(b)
        n, c = len(v), sum(cost(t.split()[0], t)[0] for t in v)
        if i < first or i > last:
            k = "wave"
        else:
            if has(b, "ds_write"):
                in_reduce = True
            if in_reduce:
                k = "reduce"
                if has(b, "global_atomic"):
                    in_reduce = False
            elif has(b, "v_exp_f32"):
                k = "quad"
            elif has(b, "v_rcp_f32"):
                k = "quad_valid"
            elif n >= 8 and all(t.split()[0].startswith("v_mov_b32") for t in v):
                k = "init"
            else:
                k = "step"
        parts[k][0] += n; parts[k][1] += c
    per = {"step": copies, "quad": 4 * copies, "quad_valid": 4 * copies, "reduce": copies, "init": max(1, sum(
        1 for b in blocks[first:last + 1] if len(valu(b)) >= 8 and all(t.split()[0].startswith("v_mov_b32") for t in valu(b)))), "wave": 1}
    import json
    out = {"INS": {k: parts[k][0] / per[k] for k in parts}, "CYC": {k: parts[k][1] / per[k] for k in parts},
           "loop_copies": copies, "source": "hipcc -S of instantsplat_amd/csrc/composite.hip, k_composite_bwd<1, false, false>; tools/isa_cost.py --bwd-model",
           "note": "init = the peeled initialisation of the nine moments: runs once per step in which the first quadrant body does not"}
    print(json.dumps(out))


def fwd_model(path):
    """Per-part costs of the forward composite kernel (k_composite_fwd<4, false>: the instantiation a 512^2 frame runs), from its
    loop structure: "step" = the walk loop's body (the one block that holds the v_exp_f32 of its two hits and branches back to
    itself), "group" = the rest of the loop over 64-record groups that contains it (staging stores, the four-quadrant-free cull of
    this wave's quadrant, prefetch, boundary store, loop control; priced as if every conditional part ran, which it does for all
    but the last group of a tile), "wave" = everything outside (prologue: unit table, first gather; epilogue: remaining boundary
    records, quadrant maximum, pixel stores)."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and "k_composite_fwdILi4ELb0ELb1" in l and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, lab = [], [], "entry"
    for l in lines[start:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((lab, cur)); cur, lab = [], m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
        if t.split()[0].startswith("s_cbranch") or t.split()[0] == "s_branch":
            blocks.append((lab, cur)); cur, lab = [], lab + "'"
    blocks.append((lab, cur))
    index = {}
    for i, (lb, _) in enumerate(blocks):
        index.setdefault(lb, i)
    walk = next(i for i, (lb, b) in enumerate(blocks) if any(t.startswith("v_exp_f32") for t in b) and b and b[-1].split()[-1] == lb)
    # the group loop: the widest backward branch around the walk block (the walk's own re-entry loop sits inside it)
    best = None
    for i, (lb, b) in enumerate(blocks):
        if i <= walk or not b or not b[-1].split()[0].startswith(("s_branch", "s_cbranch")):
            continue
        tgt = index.get(b[-1].split()[-1])
        if tgt is not None and tgt < walk and (best is None or i - tgt > best[1] - best[0]):
            best = (tgt, i)
    g0, g1 = best
    parts = {k: [0, 0] for k in ("step", "group", "wave")}
    for i, (lb, b) in enumerate(blocks):
        v = [t for t in b if t.startswith("v_")]
        k = "step" if i == walk else ("group" if g0 <= i <= g1 else "wave")
        parts[k][0] += len(v); parts[k][1] += sum(cost(t.split()[0], t)[0] for t in v)
    import json
    print(json.dumps({"INS": {k: parts[k][0] for k in parts}, "CYC": {k: parts[k][1] for k in parts}, "hits_per_step": 2,
                      "source": "hipcc -S of instantsplat_amd/csrc/composite.hip, k_composite_fwd<4, false, true>; tools/isa_cost.py --fwd-model"}))


def main():
    if sys.argv[1] == "--fwd-model":
        try:
            return fwd_model(sys.argv[2])
        except (StopIteration, ValueError, ZeroDivisionError, IndexError, OSError, TypeError) as e:
            sys.stderr.write("isa_cost.py --fwd-model: the walk / group loops were not recognised in %s (%s: %s); no table written\n"
                             % (sys.argv[2], type(e).__name__, e))
            sys.exit(3)
    if sys.argv[1] == "--bwd-model":
        try:   # pattern matching on compiler output: a different LLVM or a variant build may not have these blocks
            return bwd_model(sys.argv[2])
        except (StopIteration, ValueError, ZeroDivisionError, IndexError, OSError) as e:
            sys.stderr.write("isa_cost.py --bwd-model: the replay loop's blocks were not recognised in %s (%s: %s); "
                             "no table written\n" % (sys.argv[2], type(e).__name__, e))
            sys.exit(3)
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
