#!/usr/bin/env python3
"""Issue cost of the inner loops of a gfx950 kernel, from its assembly listing.

    hipcc --offload-arch=gfx950 <build flags> --cuda-device-only -S csrc/ses_rollout.hip -o rollout.s
    python tools/loop_issue_cost.py rollout.s k_rollout_cartpole_mlp_mix

Each VALU instruction is priced with the issue cadence measured on MI355X by tools/valu_issue.hip and
tools/vgpr_bank.hip (profiles/r01_valu_issue.txt, r01_vgpr_bank.txt), in round numbers: 2 cycles for the full-rate
kinds (mul/add/sub/mov/and/lshr/add_u32, fma forms with at most two register sources, and a three-register fma whose
source VGPRs are not all of one parity), 8 for v_rcp_f32, 4 for everything else (a three-register fma reading three
even or three odd VGPRs, min/max/med3, cvt, fract, lshl, bfi, cndmask, cmp, DPP, div_*).  The sum over a loop body is a
serial-issue estimate of one trip on a saturated SIMD -- a guide to where the cycles go, not a bound: kinds that use
different pipes overlap across waves (the LPE-4 loop runs at ~2.6 cycles per instruction at 65 536 offspring).
"""
import re
import sys

FULL = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32",
        "v_lshrrev_b32", "v_fmaak_f32", "v_fmamk_f32", "v_subrev_f32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}


def cost(line):
    parts = line.split(None, 1)
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", parts[0])
    if not op.startswith("v_"):
        return None
    if "dpp" in parts[0] or "quad_perm" in line or "row_" in line:
        return op + " (dpp)", 4
    if op in QUARTER:
        return op, 8
    if op in FULL:
        return op, 2
    if op in ("v_fma_f32", "v_fmac_f32"):
        ops = [o.strip() for o in parts[1].split(",")]
        srcs = ops[1:] if op == "v_fma_f32" else ops          # fmac reads its destination as well
        regs = {re.sub(r"[-|]", "", o) for o in srcs if re.match(r"^-?\|?[vs]\d+|^-?\|?[vs]\[", o)}
        if len(regs) <= 2:
            return op + " (<=2 reg)", 2
        vg = [int(r[1:]) for r in regs if re.match(r"^v\d+$", r)]
        if len(vg) == 3 and len({v & 1 for v in vg}) == 1:
            return op + " (3 reg, one parity)", 4
        return op + " (3 reg)", 2
    return op, 4


def loops(text, sym, min_valu=40):
    """[(mangled kernel name, [{label, valu, cycles, lds, salu, kinds}, ...])] for the first kernel whose mangled name contains
    `sym`: every loop of at least min_valu VALU instructions whose back edge targets its own header."""
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + sym + r"\w*:", l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end]
    out = []
    i = 0
    while i < len(body):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", body[i])
        if not m:
            i += 1
            continue
        label = m.group(1)
        j = next((k for k in range(i + 1, len(body))
                  if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[k])), None)
        if j is None:                                           # a loop whose back edge targets another block: skip
            i += 1
            continue
        valu = cyc = lds = salu = 0
        kinds = {}
        for l in body[i + 1:j]:
            l = l.strip()
            if l.startswith("ds_"):
                lds += 1
            elif l.startswith("s_") and not l.startswith("s_waitcnt") and not l.startswith("s_nop"):
                salu += 1
            c = cost(l) if l.startswith("v_") else None
            if c:
                valu += 1
                cyc += c[1]
                kinds[c[0]] = kinds.get(c[0], 0) + 1
        if valu >= min_valu:
            out.append({"label": label, "valu": valu, "cycles": cyc, "lds": lds, "salu": salu, "kinds": kinds})
        i = j + 1
    return text[start].split(":")[0], out


def main():
    text = open(sys.argv[1]).read().splitlines()
    name, found = loops(text, sys.argv[2])
    print(name)
    for lp in found:
        print(f"  loop {lp['label']}: {lp['valu']} VALU ({lp['cycles']} issue cycles = {lp['cycles'] / lp['valu']:.2f} per instruction), "
              f"{lp['lds']} LDS, {lp['salu']} SALU")
        print("     " + ", ".join(f"{k} x{v}" for k, v in sorted(lp["kinds"].items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
