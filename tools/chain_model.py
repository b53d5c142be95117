#!/usr/bin/env python3
"""What ONE wave alone on its SIMD needs for an env step of the fused CartPole MLP rollout -- the regime of small per-GPU
populations (512 / 1024 offspring per GPU of the strong-scaling line: fewer waves than the chip has SIMDs).

    python tools/chain_model.py profiles/r06_chain_model.json [profiles/r06_dep_latency.json]

From the gfx950 listing of `k_rollout_cartpole_mlp<16, fixed length, packed>` (the loop a 512-offspring shard runs; the scalar twin
is priced beside it) the tool builds the
register dependence graph of the loop body -- every VGPR / SGPR / vcc read and write, the loop-carried ones included -- and
prices it with the latencies tools/dep_latency.hip measured on MI355X for a lone wave (issue-to-issue time of DEPENDENT
instructions, per kind; ~3.6-3.9 ns for plain VALU, 25 ns for the table's ds_read_b128):

  chain_ns   the longest loop-carried dependence cycle: what a step costs a machine with unlimited issue.  x max_step = the
             floor of ANY lanes-per-env split of this arithmetic (more lanes per env shorten no link of it: 32 lanes per env
             ADD two; tools/time_small_populations.py measured exactly that).
  inorder_ns the same graph issued IN ORDER, one instruction per `issue_ns` (a lone wave issues an independent instruction
             every ~2.2 ns = 4-5 cycles, profiles/r01_valu_issue.txt): what this instruction order costs a lone wave.

bench.py prints chain_ns x max_step as `small_shards.small_shard_floor_us` beside the measured rollout.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import kernel_hash  # noqa: E402
from issue_model import build_flags  # noqa: E402

KERNEL = "k_rollout_cartpole_mlpILi16ELb1ELi64ELb0ELb1"      # <16 lanes per env, fixed length, 64-thread workgroups, fp32, PACKED>: what a 512-offspring shard runs


def regs_of(tok):
    """Registers named by one operand token: v12, -|v3|, v[36:39], s[2:3], vcc, exec -> ['v36', 'v37', ...]."""
    tok = tok.strip().replace("|", "").lstrip("-")
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.match(r"^[vsa]\d+$", tok):
        return [tok]
    if tok in ("vcc", "exec", "scc"):
        return [tok]
    return []


def parse(line):
    """(opcode, writes, reads, kind) of one instruction line, or None for directives / labels / waits."""
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith((".", ";")):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    if op in ("s_waitcnt", "s_nop", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_execnz", "s_cbranch_execz", "s_cbranch_vccnz",
              "s_cbranch_vccz", "s_branch", "s_endpgm"):
        return (op, [], [], "ctl")
    body = parts[1] if len(parts) > 1 else ""
    body = re.split(r"\s+(?:quad_perm|row_|op_sel|bound_ctrl|bank_mask|offset|clamp|mul:|div:)", body)[0]
    ops = [o for o in (x.strip() for x in body.split(",")) if o]
    writes, reads = [], []
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        if op.endswith("_e64"):
            writes += regs_of(ops[0])
            srcs = ops[1:]
        else:
            writes += ["vcc"]
            srcs = ops[1:] if ops and ops[0] == "vcc" else ops
        for o in srcs:
            reads += regs_of(o)
    elif base in ("v_addc_co_u32", "v_add_co_u32", "v_subb_co_u32", "v_sub_co_u32"):
        writes += regs_of(ops[0]) + regs_of(ops[1])
        for o in ops[2:]:
            reads += regs_of(o)
    elif base.startswith("ds_read"):
        writes += regs_of(ops[0])
        reads += regs_of(ops[1])
    elif base.startswith("ds_write"):
        for o in ops:
            reads += regs_of(o)
    elif base.startswith("s_cmp"):
        writes += ["scc"]
        for o in ops:
            reads += regs_of(o)
    else:
        writes += regs_of(ops[0]) if ops else []
        for o in ops[1:]:
            reads += regs_of(o)
        if base in ("v_fmac_f32", "v_mac_f32") or op.endswith("_dpp") and base in ("v_mov_b32",):
            reads += regs_of(ops[0])                       # accumulates into / keeps lanes of its destination
        if base == "v_cndmask_b32" and op.endswith("_e32"):
            reads += ["vcc"]
        if base in ("s_add_i32", "s_add_u32", "s_sub_i32"):
            writes += ["scc"]
    kind = "lds" if base.startswith("ds_read") else ("salu" if op.startswith("s_") else "valu")
    return (op, writes, reads, kind)


def latency_of(op, kind, lat):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if kind == "lds":
        return lat["lds_read"]
    if kind in ("salu", "ctl"):
        return lat["salu"]
    if base == "v_rcp_f32":
        return lat["v_rcp_f32"]
    if base.startswith("v_cmp"):
        return lat["cmp"]
    if base == "v_cndmask_b32":
        return lat["cndmask"]
    if op.endswith("_dpp"):
        return lat["dpp"]
    return lat.get(base, lat["valu"])


def loop_body(text, sym):
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*" + sym + r"\w*:", l))
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    body = text[start:end]
    best = None
    i = 0
    while i < len(body):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", body[i])
        if not m:
            i += 1
            continue
        label = m.group(1)
        j = next((k for k in range(i + 1, len(body)) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", body[k])), None)
        if j is None:
            i += 1
            continue
        ins = [p for p in (parse(l) for l in body[i + 1:j + 1]) if p]
        valu = sum(1 for p in ins if p[3] == "valu")
        ncnd = sum(1 for p in ins if p[0].startswith("v_cndmask"))
        # the step loop of fully observed envs: the long loop with the fewest selects (the masked twin has four more)
        if valu >= 70 and (best is None or ncnd < best[1]):
            best = (ins, ncnd, label, valu)
        i = j + 1
    return text[start].split(":")[0], best


def simulate(ins, lat, issue_ns, iters=8):
    """Dataflow times of `iters` trips.  Returns (period with unlimited issue, period issued in order one per issue_ns)."""
    out = []
    for in_order in (False, True):
        ready, t_issue, marks = {}, 0.0, []
        for _ in range(iters):
            last = 0.0
            for op, writes, reads, kind in ins:
                start = max([ready.get(r, 0.0) for r in reads] + [0.0])
                if in_order:
                    start = max(start, t_issue)
                    if kind in ("valu", "lds"):
                        t_issue = start + issue_ns
                done = start + latency_of(op, kind, lat)
                for w in writes:
                    ready[w] = done
                last = max(last, done)
            marks.append(last)
        out.append((marks[-1] - marks[-5]) / 4.0)
    return out


def main():
    out_path = sys.argv[1]
    dep_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_dep_latency.json")
    dep = json.load(open(dep_path))["links"]
    ns = lambda k: dep[k]["ns"]                                             # noqa: E731
    lds_link = ns("v_add_u32 + ds_read_b128 + wait + v_mov")
    pair = ns("v_cmp vcc + v_cndmask vcc")
    lat = {"valu": ns("v_fma_f32"), "v_fma_f32": ns("v_fma_f32"), "v_fmac_f32": ns("v_fmac_f32"), "v_fmamk_f32": ns("v_fmamk_f32"),
           "v_fmaak_f32": ns("v_fmamk_f32"), "v_mul_f32": ns("v_mul_f32"), "v_add_f32": ns("v_add_f32"), "v_sub_f32": ns("v_add_f32"),
           "v_min_f32": ns("v_min_f32 |x|"), "v_cvt_i32_f32": ns("v_cvt_i32_f32"), "v_fract_f32": ns("v_fract_f32"),
           "v_lshlrev_b32": ns("v_lshlrev_b32"), "v_bfi_b32": ns("v_bfi_b32"), "v_med3_f32": ns("v_med3_f32"),
           "v_rcp_f32": ns("v_rcp_f32"), "dpp": ns("v_add_f32_dpp row_ror"), "cmp": pair / 2, "cndmask": pair / 2,
           # the measured link is add + read + wait + mov: the read's own share is the link minus two plain VALU links
           "lds_read": lds_link - 2 * ns("v_lshlrev_b32"), "salu": 0.5}
    for pk, scalar in (("v_pk_fma_f32", "v_fma_f32"), ("v_pk_mul_f32", "v_mul_f32"), ("v_pk_add_f32", "v_add_f32")):
        lat[pk] = dep[pk]["ns"] if pk in dep else lat[scalar]               # (older latency files have no packed links)
    issue_ns = 2.2                                                          # profiles/r01_valu_issue.txt, one wave per SIMD
    with tempfile.TemporaryDirectory() as tmp:
        lst = os.path.join(tmp, "rollout.s")
        src = os.path.join(ROOT, "simple-es_amd", "csrc", "ses_rollout.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + build_flags() + ["--cuda-device-only", "-S", src, "-o", lst],
                              stderr=subprocess.DEVNULL)
        text = open(lst).read().splitlines()
        name, best = loop_body(text, KERNEL)
        _, twin = loop_body(text, KERNEL[:-4] + "ELb0")
    ins, _, label, valu = best
    chain, inorder = simulate(ins, lat, issue_ns)
    twin_chain, twin_inorder = simulate(twin[0], lat, issue_ns)
    lib = os.path.join(ROOT, "simple-es_amd", "libses_hip.so")
    model = {"kernel": name, "kernel_match": "k_rollout_cartpole_mlp", "loop": label, "valu_instructions": valu, "lds_reads": sum(1 for p in ins if p[3] == "lds"),
             "chain_ns": round(chain, 2), "inorder_ns": round(inorder, 2), "issue_ns": issue_ns,
             "scalar_twin": {"loop": twin[2], "valu_instructions": twin[3], "chain_ns": round(twin_chain, 2),
                             "inorder_ns": round(twin_inorder, 2)},
             "issue_only_ns": round(issue_ns * (valu + sum(1 for p in ins if p[3] == "lds")), 2),
             "kernel_code_sha256": kernel_hash.hash_kernels(lib, "k_rollout_cartpole_mlp"),
             "latencies_ns": {k: round(v, 3) for k, v in lat.items()}, "latency_source": os.path.relpath(dep_path, ROOT),
             "what": "chain_ns: longest loop-carried dependence cycle of one env step (unlimited issue) -- x max_step is the floor of any "
                     "lanes-per-env split; inorder_ns: the listing's own order, one instruction per issue_ns, operands waited for"}
    json.dump(model, open(out_path, "w"), indent=1)
    print(json.dumps({k: model[k] for k in ("loop", "valu_instructions", "chain_ns", "inorder_ns", "issue_only_ns")}))


if __name__ == "__main__":
    main()
