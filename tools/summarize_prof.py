#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into profiles/<name>.md:
kernel-trace stats of the disparity kernel + PMC counters per batched launch."""
import collections
import csv
import glob
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
root = f"gpurun_out/prof_{tag}"
lines = [f"# rocprofv3 summary `{tag}`", "", "Command: `tools/profile_gpu.sh " + tag + " " + " ".join(sys.argv[3:]) + "`", ""]
st = os.path.join(root, "trace", "trace_kernel_stats.csv")
if os.path.exists(st):
    lines += ["## `--kernel-trace --stats` (top kernels)", "", "| kernel | calls | total ns | avg ns | % | min ns | max ns |", "|---|---|---|---|---|---|---|"]
    for r in list(csv.DictReader(open(st)))[:6]:
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} | {r['MinNs']} | {r['MaxNs']} |")
tr = os.path.join(root, "trace", "trace_kernel_trace.csv")
if os.path.exists(tr):
    rows = [r for r in csv.DictReader(open(tr)) if "k_colwave" in r["Kernel_Name"] or "k_disparity_generic" in r["Kernel_Name"]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if d:
        thr = max(d) / 4
        big = [x for x in d if x > thr]
        small = [x for x in d if x <= thr]
        lines += ["", f"Disparity kernel launches: {len(d)}; batched launches (> {thr/1e3:.0f} us): n={len(big)}, "
                  f"avg {sum(big)/len(big)/1e6:.4f} ms, min {min(big)/1e6:.4f} ms; single-pair launches: n={len(small)}"
                  + (f", avg {sum(small)/len(small)/1e3:.1f} us" if small else ""),
                  f"VGPR_Count={rows[0].get('VGPR_Count')} Accum_VGPR={rows[0].get('Accum_VGPR_Count')} LDS={rows[0].get('LDS_Block_Size')} "
                  f"wg={rows[0].get('Workgroup_Size_X', rows[0].get('Workgroup_Size'))} grid={rows[0].get('Grid_Size_X', rows[0].get('Grid_Size'))}"]
lines += ["", "## PMC (separate passes; mean per batched launch of the disparity kernel)", "", "| pass | counter | mean |", "|---|---|---|"]
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "pmc_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "colwave" in r["Kernel_Name"] or "k_disparity_generic" in r["Kernel_Name"]]
    thr = max(durs) / 4 if durs else 0
    for r in rows:
        if ("colwave" in r["Kernel_Name"] or "k_disparity_generic" in r["Kernel_Name"] or r["Kernel_Name"].startswith("k_read") or r["Kernel_Name"].startswith("k_write")):
            if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > thr or not ("colwave" in r["Kernel_Name"]):
                agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        lines.append(f"| {os.path.basename(os.path.dirname(f))[4:30]} | {c} ({k}) | {sum(v)/len(v):.6g} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
