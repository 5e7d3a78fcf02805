#!/usr/bin/env python3
"""The WHOLE parameter grid of the reference's UI (cmd/handlers/params.go:37-52: BlockSize 3..31 odd, MaxDisparity 16..256
step 16) plus the defaults' BlockSize 16, at 1920x1080 with 64 device-resident pairs per launch, in ONE process: per cell the
plan, the kernel kind that ran, microseconds per pair and parity of frame 0 on two row bands against the literal restatement
(multi-threaded).  --real: frame 0 is the reference's own 1920x1080 pair (testdata/im0 + im1 through the Go gray formula) and six
bands of it are checked per cell.  usage (GPU box): tools/survey_grid_full.py [out.txt] [pairs] [--real]   -> also prints a markdown table."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _pinned import to_device, to_host
import oracle
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi

args = [a for a in sys.argv[1:] if not a.startswith("--")]
REAL = "--real" in sys.argv          # frame 0 = the reference's own 1080p pair (tests/golden/im0.png + im1.png), more bands checked
out_path = args[0] if len(args) > 0 else os.path.join(REPO, "gpurun_out", "grid_full.txt")
N = int(args[1]) if len(args) > 1 else 64
W, H = 1920, 1080
dev = torch.device("cuda", 0)
parts = [sha.synth.synth_batch_torch([1000 + i for i in range(c, min(c + 32, N))], W, H, dev) for c in range(0, N, 32)]
Ls, Rs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]); del parts
if REAL:
    from PIL import Image
    g = os.path.join(REPO, "tests", "golden")
    Ls[0] = to_device(oracle.gray_from_rgb8(np.array(Image.open(os.path.join(g, "im0.png")))), dev)
    Rs[0] = to_device(oracle.gray_from_rgb8(np.array(Image.open(os.path.join(g, "im1.png")))), dev)
out = torch.empty_like(Ls)
ctx = capi.Context(0, W, H, 1)
L0, R0 = to_host(Ls[0]), to_host(Rs[0])
BANDS = (0, 137, 402, H // 2 - 1, 803, H - 3) if REAL else (0, H // 2 - 1)
thr = min(16, os.cpu_count() or 1)
blocks = [3, 5, 7, 9, 11, 13, 15, 16, 17, 19, 21, 23, 25, 27, 29, 31]
disps = list(range(16, 257, 16))
cells, bad, t_start = {}, 0, time.time()
with open(out_path, "w") as f:
    for b in blocks:
        for d in disps:
            plan = capi.plan_describe(W, H, b, d)
            for _ in range(2):
                ctx.disparity_dev(Ls, Rs, out, b, d)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ctx.disparity_dev(Ls, Rs, out, b, d)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 / N * 1e3
            kind = capi.last_kernel_kind()
            o0 = to_host(out[0])
            ok = True
            for y0 in BANDS:
                ok = ok and np.array_equal(o0[y0:y0 + 3], oracle.disparity_rows_mt(L0, R0, b, d, y0, y0 + 3, thr))
            bad += 0 if ok else 1
            shape = "+".join(f"{p['nw'] - p['nwh']}x{p['nd']}" + (f"+{p['nwh']}x{p['nd'] - 4 if p['packed'] else p['nd'] // 2}" if p["nwh"] else "") + ("+1" if p["extra"] else "") for p in plan)
            exact = all((p["nw"] - p["nwh"]) * p["nd"] + p["nwh"] * (p["nd"] - 4 if p["packed"] else p["nd"] // 2) + p["extra"] == p["count"] for p in plan)
            line = f"{W}x{H} b{b} D{d} n={N} kind={kind}: {us:.1f} us/frame  plan {shape}  exact={exact}  parity={ok}"
            print(line, flush=True); f.write(line + "\n")
            cells[(b, d)] = us
    f.write(f"# {len(cells)} cells, {bad} parity failures, {time.time() - t_start:.0f} s\n")
print("| BlockSize \\ MaxDisparity | " + " | ".join(str(d) for d in disps) + " |")
print("|---|" + "---|" * len(disps))
for b in blocks:
    print(f"| {b}{' (default)' if b == 16 else ''} | " + " | ".join(f"{cells[(b, d)]:.1f}" for d in disps) + " |")
ctx.close()
sys.exit(1 if bad else 0)
