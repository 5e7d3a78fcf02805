#!/usr/bin/env python3
"""After tools/make_profiles.sh <tag> ran on the GPU box and gpurun merged gpurun_out/ back:
copies the summaries into profiles/ (tracked) and writes the two JSON files bench.py reads,
each stamped with the digest of the kernel sources they were measured on:
  profiles/traffic.json      HBM bytes (FETCH_SIZE x2 + WRITE_SIZE) and SQ_INSTS_VALU per batched c3 launch
  profiles/valu_budget.json  per-wave-step instruction-class budget of the c3 kernel (tools/valu_budget.py)
usage: tools/collect_profiles.py <tag>"""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import valu_budget  # noqa: E402

tag = sys.argv[1]
go, pr = os.path.join(REPO, "gpurun_out"), os.path.join(REPO, "profiles")
os.makedirs(pr, exist_ok=True)
for name in sorted(os.listdir(go)):
    if name.startswith(tag + "_") and (name.endswith(".md") or name == tag + "_bench_line.json"):    # summaries + the bench line, not scratch outputs
        shutil.copy(os.path.join(go, name), os.path.join(pr, name))
        print("profiles/" + name)
st = os.path.join(go, f"prof_{tag}_bench", "trace", "trace_kernel_stats.csv")
if os.path.exists(st):
    rows = [r for r in csv.DictReader(open(st))][:8]
    with open(os.path.join(pr, f"{tag}_bench_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:110]
            w.writerow(r)
    print(f"profiles/{tag}_bench_kernel_stats.csv")
digest = valu_budget.source_digest()
md = open(os.path.join(go, f"{tag}_c3.md")).read()


def num(pat):
    m = re.search(pat, md)
    return float(m.group(1)) if m else None


frames = int(num(r"Dmax \d+, (\d+) device-resident pairs per launch") or 0)
fetch = num(r"\| pmc_fetch \| FETCH_SIZE \| ([\d.e+]+) \|")
write = num(r"\| pmc_write \| WRITE_SIZE \| ([\d.e+]+) \|")
insts = num(r"\| pmc_sq \| SQ_INSTS_VALU \| ([\d.e+]+) \|")
tj = {"_comment": "HBM traffic and VALU instruction count of one batched launch of the disparity kernel at the bench.py shape, "
                  "from rocprofv3 --pmc (separate passes; profiles/%s_c3.md). gfx950 correction per MI355X_MICROARCH.md "
                  "(FETCH_SIZE x2, WRITE_SIZE x1; re-calibrated in round 1 on known-byte kernels, tools/ubench/hbm_calib.hip)." % tag,
      "round": int(re.sub(r"\D", "", tag) or 0), "source": f"profiles/{tag}_c3.md", "source_digest": digest,
      "frames_per_launch": frames, "width": 1920, "height": 1080, "block": 9, "maxdisp": 64,
      "fetch_size_kb": fetch, "write_size_kb": write, "fetch_correction": 2.0, "write_correction": 1.0,
      "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024), "sq_insts_valu_per_launch": int(insts)}
json.dump(tj, open(os.path.join(pr, "traffic.json"), "w"), indent=1)
print("profiles/traffic.json", tj["hbm_bytes_per_launch"], tj["sq_insts_valu_per_launch"], digest)
plan = subprocess.check_output([sys.executable, "-c",
                                "import sys; sys.path.insert(0, %r); import steroscopic_hardware_amd as s; print(s.capi.plan_describe(1920,1080,9,64))" % REPO], text=True)
pl = eval(plan.strip())[0]
sys.argv = ["valu_budget.py", str(pl["p"]), str(pl["nd"]), "4", str(pl["packed"]), "1024", str(pl["w16"]), "1" if (pl["p"] == 4 and pl["packed"]) else "0", "--json", os.path.join(pr, "valu_budget.json")]
valu_budget.main()
