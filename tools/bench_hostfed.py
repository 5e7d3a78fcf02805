#!/usr/bin/env python3
"""Host-fed (PCIe-inclusive) rate of the C-ABI host entry points at 1920x1080 b9 D64."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import steroscopic_hardware_amd as sha
W, H, B, D, N = 1920, 1080, 9, 64, 32
pairs = [sha.synth.synth_pair(1000 + i, W, H) for i in range(4)]
Ls = [pairs[i % 4][0] for i in range(N)]; Rs = [pairs[i % 4][1] for i in range(N)]
ctx = sha.capi.Context(0, W, H, 1)
ctx.disparity(Ls[0], Rs[0], B, D)
t0 = time.perf_counter()
o1 = np.zeros((H, W), np.uint8)
for i in range(8): ctx.disparity(Ls[i], Rs[i], B, D, out=o1)
t1 = time.perf_counter()
k, t = ctx.last_timing()
print(f"sadhip_disparity_u8 (one pair, host buffers, synchronous): {(t1-t0)/8*1e3:.3f} ms/pair wall; last call kernel {k:.3f} ms, total {t:.3f} ms")
outs = [np.zeros((H, W), np.uint8) for _ in range(N)]          # pre-touched output pages
ctx.disparity_batch(Ls, Rs, B, D, outs=outs)
t0 = time.perf_counter(); ctx.disparity_batch(Ls, Rs, B, D, outs=outs); t1 = time.perf_counter()
print(f"sadhip_disparity_batch_u8 ({N} pairs, double-buffered H2D/kernel/D2H): {(t1-t0)/N*1e3:.3f} ms/pair = {N/(t1-t0):.0f} pairs/s (PCIe-inclusive)")
