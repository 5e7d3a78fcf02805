#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_cfg.sh) into a markdown summary for profiles/.

  tools/summarize_cfg.py <tag> <out.md> W H BLOCK MAXDISP NFRAMES

Reports, for the batched launches of the disparity kernel (the longest dispatches of the run):
kernel-trace duration and resources (VGPRs, LDS, scratch of THOSE dispatches), the PMC counters per
launch, and the derived figures DESIGN.md quotes: effective clock (GRBM_GUI_ACTIVE / 8 / duration),
VALU wave-instructions per 64 candidate evaluations, SIMD-cycles per VALU instruction, HBM traffic
(FETCH_SIZE x2 + WRITE_SIZE, KB -> bytes) against the algorithmic 3*W*H*N bytes."""
import collections
import csv
import os
import re
import sys

tag, out = sys.argv[1], sys.argv[2]
W, H, B, D, N = (int(v) for v in sys.argv[3:8])
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"prof_{tag}")
KERNELS = ("k_colwave", "k_disparity_generic", "k_fcol", "k_mcol")


def is_disp(name):
    return any(k in name for k in KERNELS)


def cands(w, h, b, d):
    hh = b // 2
    return h * (hh + d * (d + 1) // 2 + (w - hh - d) * (d + 1)) if w - hh - d > 0 else None


lines = [f"# rocprofv3 summary `{tag}`: {W}x{H} block {B} Dmax {D}, {N} device-resident pairs per launch", "",
         f"Command (on the MI355X box): `tools/profile_cfg.sh {tag} {W} {H} {B} {D} {N}` "
         "(kernel trace, then three separate `--pmc` passes over `tools/bench_cfg.py`)", ""]
log = os.path.join(root, "trace.log")
if os.path.exists(log):
    last = [l for l in open(log).read().splitlines() if "us/frame" in l]
    if last:
        lines += ["bench_cfg.py under the kernel trace: `" + last[-1].strip() + "`", ""]
tr = os.path.join(root, "trace", "trace_kernel_trace.csv")
big_ns = None
if os.path.exists(tr):
    rows = [r for r in csv.DictReader(open(tr)) if is_disp(r["Kernel_Name"])]
    for r in rows:
        r["_d"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if rows:
        thr = max(r["_d"] for r in rows) / 50        # (a job may be two kernels of very different size: the BlockSize-9 tile split)
        big = [r for r in rows if r["_d"] > thr]
        by_kernel = collections.defaultdict(list)
        for r in big:
            by_kernel[re.sub(r"^.*(k_\w+<[^>]*>).*$", r"\1", r["Kernel_Name"])[:70]].append(r)
        lines += ["## `--kernel-trace` (batched launches of the disparity kernel)", "",
                  "(The trace's VGPR / LDS columns are what rocprofv3 prints for the dispatch packet: LDS is 0 because the kernel's LDS is "
                  "dynamic, and its VGPR figure is the allocation granule, not the compiler's count -- see `hipcc -Rpass-analysis=kernel-resource-usage` "
                  "in `steroscopic-hardware_amd/lib/kernels_fast.remarks` / DESIGN.md for registers, scratch and LDS per variant.)", "",
                  "| kernel | launches | avg us | min us | trace VGPR field | trace LDS field | scratch | workgroup | grid (threads) |", "|---|---|---|---|---|---|---|---|---|"]
        tot = 0.0
        for k, rs in by_kernel.items():
            d = [r["_d"] for r in rs]
            r0 = rs[0]
            lines.append(f"| `{k}` | {len(d)} | {sum(d)/len(d)/1e3:.1f} | {min(d)/1e3:.1f} | {r0['VGPR_Count']} | {r0['LDS_Block_Size']} | "
                         f"{r0['Scratch_Size']} | {r0['Workgroup_Size_X']} | {r0['Grid_Size_X']}x{r0['Grid_Size_Y']}x{r0['Grid_Size_Z']} |")
            tot += sum(d) / len(d)
        big_ns = tot
        lines += ["", f"Per pair: {tot / N / 1e3:.2f} us (sum over the kernels of one job / {N} pairs)."]
        c = cands(W, H, B, D)
        if c:
            lines += [f"Candidate evaluations per pair {c:,} -> {c * N / tot:.0f} Gcand/s under the profiler."]
st = os.path.join(root, "trace", "trace_kernel_stats.csv")
if os.path.exists(st):
    lines += ["", "`--stats` rows of the disparity kernels:", "", "| kernel | calls | total ns | avg ns | min ns | max ns |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(st)):
        if is_disp(r["Name"]):
            kname = re.sub(r"^.*(k_[a-z_]+<[^>]*>).*$", r"\1", r["Name"])[:70]
            lines.append(f"| `{kname}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |")

pm = {}
for p in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = os.path.join(root, p, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if is_disp(r["Kernel_Name"])]
    if not rows:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in rows:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    thr = max(dur.values()) / 50
    # one job may be several kernels (passes): sum the counters of the big dispatches, divide by jobs
    bigs = [d for d in per if dur[d] > thr]
    names = collections.Counter(r["Kernel_Name"] for r in rows if dur[r["Dispatch_Id"]] > thr and r["Counter_Name"] == rows[0]["Counter_Name"])
    jobs = min(names.values()) if names else 1
    acc = collections.defaultdict(float)
    for d in bigs:
        for c, v in per[d].items():
            acc[c] += v
    pm[p] = {c: v / jobs for c, v in acc.items()}
    pm[p]["_ns"] = sum(dur[d] for d in bigs) / jobs
if pm:
    lines += ["", "## PMC (separate passes; per job = all kernels of one batched call)", "", "| pass | counter | per job |", "|---|---|---|"]
    for p, cs in pm.items():
        for c, v in sorted(cs.items()):
            if not c.startswith("_"):
                lines.append(f"| {p} | {c} | {v:.6g} |")
        lines.append(f"| {p} | (kernel duration in this pass, us) | {cs['_ns'] / 1e3:.1f} |")
    lines += ["", "## Derived", ""]
    sq = pm.get("pmc_sq")
    if sq and "GRBM_GUI_ACTIVE" in sq:
        cyc = sq["GRBM_GUI_ACTIVE"] / 8
        clk = cyc / sq["_ns"]
        lines.append(f"* effective clock in the SQ pass: GRBM_GUI_ACTIVE / 8 / duration = {clk:.2f} GHz")
        if "SQ_INSTS_VALU" in sq:
            simd_cyc = cyc * 1024
            lines.append(f"* VALU wave-instructions per job {sq['SQ_INSTS_VALU']:.4g}; SIMD-cycles available {simd_cyc:.4g} "
                         f"-> {simd_cyc / sq['SQ_INSTS_VALU']:.2f} SIMD-cycles per VALU instruction")
            c = cands(W, H, B, D)
            if c:
                slots = c * N / 64
                lines.append(f"* per 64 candidate evaluations (one wave-wide slot): {sq['SQ_INSTS_VALU'] / slots:.2f} VALU instructions, "
                             f"{simd_cyc / slots:.1f} SIMD-cycles")
        if "SQ_WAVES" in sq:
            lines.append(f"* waves per job {sq['SQ_WAVES']:.0f}")
    ft, wr = pm.get("pmc_fetch"), pm.get("pmc_write")
    if ft and wr and "FETCH_SIZE" in ft and "WRITE_SIZE" in wr:
        traffic = ft["FETCH_SIZE"] * 1024 * 2 + wr["WRITE_SIZE"] * 1024
        alg = 3 * W * H * N
        lines.append(f"* HBM traffic per job: FETCH_SIZE {ft['FETCH_SIZE']:.0f} KB x2 (gfx950 correction) + WRITE_SIZE {wr['WRITE_SIZE']:.0f} KB "
                     f"= {traffic / 1e6:.1f} MB; algorithmic 3*W*H*N = {alg / 1e6:.1f} MB; ratio {traffic / alg:.2f}")
        if big_ns:
            lines.append(f"* algorithmic bytes / trace duration = {alg / big_ns:.1f} GB/s = {alg / big_ns / 8000:.4f} of 8 TB/s")
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
