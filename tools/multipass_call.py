#!/usr/bin/env python3
"""Round 5: what a multi-pass call (more than 129 candidates: a u32 key per pixel travels between the passes in stream-ordered
scratch, kernels_generic.hip scratch_alloc) costs around its kernels.  One device-resident 1080p pair per call, launch +
synchronise, median wall time and the kernels' own time (events) -- single-pass neighbours for comparison; then 64 pairs.
usage (GPU box): tools/multipass_call.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
if os.environ.get("SADHIP_LIB"): capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])
W, H = 1920, 1080
Ls, Rs = sha.synth.synth_batch_torch([1000 + i for i in range(64)], W, H, "cuda:0")
out = torch.empty_like(Ls)
ctx = capi.Context(0, W, H, 1)
for n in (1, 64):
    for B, D in ((16, 128), (16, 144), (16, 192), (16, 256), (9, 128), (9, 144), (9, 256), (31, 256)):
        passes = len(capi.plan_describe(W, H, B, D))
        for _ in range(3): ctx.disparity_dev(Ls[:n], Rs[:n], out[:n], B, D)
        torch.cuda.synchronize()
        walls, kerns = [], []
        for _ in range(40 if n == 1 else 8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(); ctx.disparity_dev(Ls[:n], Rs[:n], out[:n], B, D); e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0); kerns.append(e0.elapsed_time(e1)); enq = t1 - t0
        walls.sort(); kerns.sort()
        print(f"n={n:2d} B{B} D{D} passes={passes}: wall {walls[len(walls)//2]*1e3:.3f} ms  events {kerns[len(kerns)//2]:.3f} ms  per pair {kerns[len(kerns)//2]/n*1e3:.1f} us  (last enqueue {enq*1e3:.3f} ms)", flush=True)
ctx.close()
