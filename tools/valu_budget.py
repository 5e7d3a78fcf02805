#!/usr/bin/env python3
"""Per-wave-step instruction budget of a column-wave kernel instantiation, from the compiler's
assembly (hipcc -S of csrc/kernels_fast.hip), by issue class.

  tools/valu_budget.py [P ND H PK WB W16 [TR]] [--asm lib/kernels_fast.s] [--json out.json] [--blocks]
  (TR = 1: the triple-column form of the BlockSize-9 kernels, the one that runs all tiles but the image's last)

The row loop is the largest natural loop of the kernel (back edge with the widest span).  Every
instruction in it is put into one class; the class costs are the ones measured on MI355X by
tools/ubench/valu_rates.hip and valu_rates2.hip (SIMD-cycles per wave-instruction with >= 2 waves
per SIMD):

  fast   2   v_add/sub/subrev_u32, v_and/or/xor_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, the u16 / f32 /
             f16 add/sub/mul/fma/min/max of VOP2 (u16/i16 min/max only), v_cmp_* -- with VGPR / inline-constant
             operands only (measured 2.2-2.6)
  slow   4   everything else on the vector ALU: v_sad_*, v_min/max_u32/i32/f32, v_min3/max3/med3, v_perm,
             v_lshl_or / v_lshlrev / v_and_or / v_bfi / v_add3 / v_mad_*, v_cndmask, v_cvt_*, every v_pk_*,
             every DPP or SDWA form, and a fast-class opcode reading an SGPR or a 32-bit literal (measured 4.1-4.5)
  qrate 16   v_qsad_pk_u16_u8, v_mqsad_* (measured 16.2)
  nop    4   s_nop (costs a vector issue slot here, tools/ubench/block_bench.hip)
  other  0   LDS, vector memory, scalar ALU / memory, branches, waits (their own issue ports)

Output: counts per class for the whole loop and (with --blocks) per basic block, the loop's ideal
SIMD-cycles at those costs, and the mix ceiling = 1024 SIMDs x 2.4 GHz / (cycles per VALU instruction).
The blocks of the extra-candidate path (one wave of the workgroup only) are listed separately when the
kernel has them (they contain v_sad_u32 in a packed kernel).
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "steroscopic-hardware_amd", "csrc")
LIBD = os.path.join(REPO, "steroscopic-hardware_amd", "lib")

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_u16", "v_sub_u16", "v_min_u16", "v_max_u16", "v_min_i16", "v_max_i16", "v_add_f32", "v_sub_f32",
        "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_add_f16", "v_sub_f16", "v_not_b32"}
COST = {"fast": 2, "slow": 4, "qrate": 16, "nop": 4, "other": 0}


def source_digest():
    """digest of the kernel sources (everything under csrc/ except sadhip.hip, the host-side C ABI)"""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inc")) and f != "sadhip.hip":
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def ensure_asm(path, defs=()):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    if os.path.exists(path) and os.path.getmtime(path) >= max(os.path.getmtime(f) for f in srcs):
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-command-line-argument",
                           "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", path, os.path.join(CSRC, "kernels_fast.hip")] + ["-D" + d for d in defs])


def classify(line):
    ins = line.split(";")[0].strip()
    if not ins:
        return None, None
    op = ins.split()[0]
    if op == "s_nop":
        return op, "nop"
    if not op.startswith("v_"):
        return op, "other"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith(("v_qsad", "v_mqsad")):
        return op, "qrate"
    if op.endswith(("_dpp", "_sdwa")) or " row_" in ins or " wave_" in ins or "quad_perm" in ins or "dst_sel" in ins:
        return op, "slow"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        cls = "fast"
    elif base in FAST:
        cls = "fast"
    else:
        return op, "slow"
    ops = ins[len(op):]
    # scalar register, vcc/exec, or a 32-bit literal operand makes a fast opcode slow
    srcs = [o.strip() for o in ops.split(",")][1:]
    if base.startswith("v_cmp"):
        srcs = [o.strip() for o in ops.split(",")]
        srcs = [o for o in srcs if not re.match(r"^(vcc|s\[\d+:\d+\])$", o)]      # the mask destination is free
    for o in srcs:
        if re.match(r"^s\d+$|^s\[|^vcc|^exec|^m0$|^ttmp", o):
            return op, "slow"
        m = re.match(r"^(-?0x[0-9a-fA-F]+|-?\d+)$", o)
        if m:
            v = int(m.group(1), 0)
            if not (-16 <= v <= 64):
                return op, "slow"
    return op, cls


def kernel_body(asm, sym, form=None):
    """The kernel's assembly lines; form (1 plain, 3 triple-column): only the part from that form's row-loop
    marker to the next marker -- the BlockSize-9 kernels hold both forms of the row loop."""
    i = asm.index("\n" + sym + ":")
    j = asm.index(".Lfunc_end", i)
    lines = asm[i:j].splitlines()
    marks = [k for k, ln in enumerate(lines) if "SADHIP_ROWLOOP form=" in ln]
    if form is not None and len(marks) > 1:
        for n, k in enumerate(marks):
            if f"form={form} " in lines[k]:
                return lines[k:marks[n + 1] if n + 1 < len(marks) else len(lines)]
    return lines


def analyse(lines):
    labels, ins = {}, []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if "SADHIP_XTRA_BEGIN" in ln or "SADHIP_XTRA_END" in ln:
            ins.append(";" + ("XB" if "BEGIN" in ln else "XE"))       # markers the kernel emits around the extra-candidate path
            continue
        if "SADHIP_MASKED_BEGIN" in ln or "SADHIP_MASKED_END" in ln:
            ins.append(";" + ("MB" if "BEGIN" in ln else "ME"))       # ... and around the key-masking reduce of the leftmost tiles
            continue
        if ln.startswith("\t") and not ln.lstrip().startswith((".", ";")):
            ins.append(ln.strip())
    best = None
    for k, t in enumerate(ins):
        m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            span = (labels[m.group(1)], k)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        raise SystemExit("no loop found")
    inv = sorted((v, k) for k, v in labels.items())
    blocks = []
    cur = {"label": "(loop head)", "ins": []}
    starts = {v: k for v, k in inv}
    for idx in range(best[0], best[1] + 1):
        if idx in starts and cur["ins"]:
            blocks.append(cur)
            cur = {"label": starts[idx], "ins": []}
        elif idx in starts:
            cur["label"] = starts[idx]
        cur["ins"].append(ins[idx])
    blocks.append(cur)
    return blocks


def summarise(inss):
    cnt = {k: 0 for k in COST}
    ops = {}
    for t in inss:
        op, cls = classify(t)
        if op is None:
            continue
        cnt[cls] += 1
        if cls != "other":
            ops[(cls, op)] = ops.get((cls, op), 0) + 1
    return cnt, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variant", nargs="*", default=["4", "16", "4", "1", "1024", "0", "1"], help="P ND H PK WB W16 [TR]")
    ap.add_argument("--asm", default=os.path.join(LIBD, "kernels_fast.s"))
    ap.add_argument("--json")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("-D", dest="defs", action="append", default=[], help="extra -D for the compile")
    a = ap.parse_args()
    P, ND, H, PK, WB, W16 = (int(v) for v in a.variant[:6])
    TR = int(a.variant[6]) if len(a.variant) > 6 else 0
    ensure_asm(a.asm, a.defs)
    asm = open(a.asm).read()
    sym = f"_ZN6sadhip12_GLOBAL__N_19k_colwaveILi{P}ELi{ND}ELi{H}ELb{PK}ELi{WB}ELb{W16}ELb{TR}ELb0EEEvNS0_6CwArgsE"   # k_colwave<P, ND, H, PK, WB, W16, TR, MIX = false>
    blocks = analyse(kernel_body(asm, sym, 3 if TR else 1))
    main_ins, xtra_ins = [], []
    # the extra-candidate path = the blocks from the kernel's SADHIP_XTRA_BEGIN marker to its SADHIP_XTRA_END
    for b in blocks:
        b["xtra"] = False
    xb = [i for i, b in enumerate(blocks) if ";XB" in b["ins"]]
    xe = [i for i, b in enumerate(blocks) if ";XE" in b["ins"]]
    if xb and xe:
        for i in range(xb[0], xe[-1] + 1):
            blocks[i]["xtra"] = True
    mb = [i for i, b in enumerate(blocks) if ";MB" in b["ins"]]
    me = [i for i, b in enumerate(blocks) if ";ME" in b["ins"]]
    masked_ins = []
    for b in blocks:
        b["masked"] = False
    if mb and me:
        for i in range(mb[0], me[-1] + 1):
            blocks[i]["masked"] = True
    for b in blocks:
        (xtra_ins if b["xtra"] else masked_ins if b["masked"] else main_ins).extend(b["ins"])
    cnt, ops = summarise(main_ins)
    xcnt, _ = summarise(xtra_ins)
    mcnt, _ = summarise(masked_ins)
    valu = cnt["fast"] + cnt["slow"] + cnt["qrate"]
    cyc = sum(cnt[k] * COST[k] for k in COST)
    slots = P * ND * (1 if not PK else 1)          # (column, candidate) pairs per lane per step = wave-wide slots of 64 candidates
    res = {
        "kernel": f"k_colwave<{P},{ND},{H},{'true' if PK else 'false'},{WB},{'true' if W16 else 'false'}{',triple-column' if TR else ''}>",
        "source_digest": source_digest(),
        "class_cost_cycles": COST,
        "row_loop_common_path": {"counts": cnt, "valu_instructions": valu, "ideal_simd_cycles": cyc,
                                 "cycles_per_valu_instruction": cyc / max(valu, 1)},
        "extra_candidate_path_one_wave_only": {"counts": xcnt, "ideal_simd_cycles": sum(xcnt[k] * COST[k] for k in COST)},
        "masked_reduce_leftmost_tiles_only": {"counts": mcnt, "ideal_simd_cycles": sum(mcnt[k] * COST[k] for k in COST)},
        "slots_per_wave_step": slots,
        "valu_per_slot": valu / slots, "ideal_cycles_per_slot": cyc / slots,
        "mix_ceiling_Gwinst_per_s": 1024 * 2.4 / (cyc / max(valu + cnt["nop"], 1)),
        "vector_peak_2cycle_Gwinst_per_s": 1024 * 2.4 / 2,
        "by_opcode": {f"{c}:{o}": n for (c, o), n in sorted(ops.items(), key=lambda x: (-x[1]))},
    }
    print(f"{res['kernel']}  (source digest {res['source_digest']})")
    print(f"row loop, common path: {valu} VALU ({cnt['fast']} fast, {cnt['slow']} slow, {cnt['qrate']} quarter-rate) + {cnt['nop']} s_nop"
          f" + {cnt['other']} other; ideal {cyc} SIMD-cycles = {cyc / max(valu, 1):.2f} per VALU instruction")
    print(f"extra-candidate path (one wave per workgroup): {xcnt['fast']} fast, {xcnt['slow']} slow, {xcnt['nop']} s_nop, {xcnt['other']} other")
    if masked_ins:
        print(f"key-masking reduce (waves of the leftmost tile, instead of the plain reduce): {mcnt['fast']} fast, {mcnt['slow']} slow, {mcnt['nop']} s_nop")
    print(f"per slot of 64 candidates ({slots} slots per wave-step): {valu / slots:.2f} VALU, {cyc / slots:.2f} ideal SIMD-cycles")
    print(f"mix ceiling {res['mix_ceiling_Gwinst_per_s']:.1f} G wave-instructions/s; 2-cycle vector peak {res['vector_peak_2cycle_Gwinst_per_s']:.1f}")
    print("by opcode:", ", ".join(f"{k} x{v}" for k, v in list(res["by_opcode"].items())[:24]))
    if a.blocks:
        for b in blocks:
            c, _ = summarise(b["ins"])
            print(f"  {b['label']:>12} {'xtra' if b['xtra'] else 'mask' if b['masked'] else '    '} fast {c['fast']:4d} slow {c['slow']:4d} nop {c['nop']:3d} other {c['other']:4d}")
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
    return res


if __name__ == "__main__":
    main()
