#!/usr/bin/env python3
"""Round 5: one synchronous call per pair at the reference's own frame size (testdata/: 640x480) and neighbours, from pageable and
from page-locked memory, gray and RGBA (device / host gray): median wall time, the kernel's share.  SADHIP_TUNE_PAIRUP=0 (diagnostics
build) = the two planes of a small pair as two uploads, for the A/B.  usage (GPU box): tools/small_call.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
if any(k.startswith("SADHIP_TUNE_") for k in os.environ): capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libsadhip_tune.so")
def med(fn, n=64):
    for k in range(6): fn(k)
    ts = []
    for k in range(n):
        t0 = time.perf_counter(); fn(k); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3
rng = np.random.default_rng(3)
for (w, h, b, d) in ((640, 480, 16, 64), (640, 480, 7, 32), (320, 240, 16, 64), (960, 540, 16, 64), (1280, 720, 9, 64)):
    with capi.Context(0, w, h, 1) as c, capi.HostBuffer(3 * 4 * w * h) as hb:
        pairs = [sha.synth.synth_pair(70 + k, w, h) for k in range(4)]
        pl = hb.view((3, 4, h, w))
        for k in range(4): pl[0, k], pl[1, k] = pairs[k]
        pics = rng.integers(0, 256, (2, 2, h, w, 4), dtype=np.uint8); pics[..., 3] = 255
        a = med(lambda k: c.disparity(pairs[k & 3][0], pairs[k & 3][1], b, d))
        kern = c.last_timing()[0]
        p = med(lambda k: c.disparity(pl[0, k & 3], pl[1, k & 3], b, d, out=pl[2, k & 3]))
        r1 = med(lambda k: c.disparity_rgba(pics[k & 1][0], pics[k & 1][1], b, d))
        r2 = med(lambda k: c.disparity_rgba(pics[k & 1][0], pics[k & 1][1], b, d, capi.FLAG_HOST_GRAY))
        print(f"{w}x{h} b{b} D{d}: pageable gray {a:.3f} ms, page-locked gray {p:.3f}, RGBA device gray {r1:.3f}, RGBA host gray {r2:.3f}; kernel {kern:.3f}", flush=True)
