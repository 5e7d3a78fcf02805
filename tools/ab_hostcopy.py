#!/usr/bin/env python3
"""What staging pageable caller memory through the context's page-locked planes costs, how many bands the staged copies
should be cut into, and (round 5) what the host-side gray conversion of SADHIP_FLAG_HOST_GRAY and the copy threads' spin
budget change.  The thread / band sweeps need the diagnostics build (SADHIP_LIB=.../libsadhip_tune.so; the context reads
SADHIP_TUNE_BANDKB / SADHIP_TUNE_COPYTHREADS when it is created); --quick, --spin and the default line run on the product
library.  1920x1080, BlockSize 9, MaxDisparity 64 (BASELINE config 3): median of 30 synchronous sadhip_disparity_u8 calls on
pageable numpy arrays, a 64-pair sadhip_disparity_batch_u8 on pageable arrays, sadhip_disparity_rgba_u8 (device gray, host
gray on opaque pictures, host gray on pictures with arbitrary alpha) and sadhip_gray_u8 on pageable RGBA pictures.
(Rounds 2-4's in-place path -- the runtime locks the caller's pages -- is in no build any more; its numbers are the "direct"
line of profiles/r04_ab_hostcopy.txt.)
usage (GPU box): [SADHIP_LIB=steroscopic-hardware_amd/lib/libsadhip_tune.so] tools/ab_hostcopy.py [--quick | --spin | ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
if os.environ.get("SADHIP_LIB"): capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])
W, H, B, D, NB = 1920, 1080, 9, 64, 64
if "--pin" in sys.argv:                                       # the calling thread (before it allocates the frames; the copy threads inherit the mask) on the device's own CPUs
    place = capi.device_describe(0)
    if place["cpus"]: os.sched_setaffinity(0, place["cpus"])
    print("pinned to", len(place["cpus"]), "CPUs of NUMA node", place["numa_node"], flush=True)
pairs = [sha.synth.synth_pair(100 + i, W, H) for i in range(NB)]
ls, rs = [p[0] for p in pairs], [p[1] for p in pairs]
os_ = [np.zeros((H, W), np.uint8) for _ in range(NB)]
rgba_any = np.random.default_rng(5).integers(0, 256, (2, H, W, 4), dtype=np.uint8)     # arbitrary alpha: the general path of the host conversion
rgba = rgba_any.copy(); rgba[..., 3] = 255                                              # opaque, what a camera picture is
ref = None

def med(fn, n=30):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3, min(ts) * 1e3

if "--pairup" in sys.argv:                                    # a staged pair of a batch as one upload (default) or as two
    modes = [("default", 0, 0), ("twouploads", 0, 0), ("default", 0, 0), ("twouploads", 0, 0)]
elif "--dirstreams" in sys.argv:                                # staged batch on per-direction streams (default) or one stream per slot
    modes = [("default", 0, 0), ("slotstreams", 0, 0), ("default", 0, 0), ("slotstreams", 0, 0)]
elif "--stream" in sys.argv:                                    # non-temporal stores in the staged copies on / off (SADHIP_TUNE_STREAM)
    modes = [("default", 0, 0), ("nostream", 0, 0), ("default", 0, 0), ("nostream", 0, 0)]
elif "--quick" in sys.argv:
    modes = [("default", 0, 0), ("default", 0, 0)]
elif "--spin" in sys.argv:                                      # sadhip_ctx_set_copy_spin_us: the default 150 us against sleeping at once
    modes = [("default", 0, 0), ("spin0", 0, 0), ("spin50", 0, 0), ("default", 0, 0), ("spin0", 0, 0), ("spin50", 0, 0)]
else:
  modes = [("default", 0, 0)] + [("staged", 8192, t) for t in (1, 2, 4, 6, 8)] + [("staged", kb, 6) for kb in (2048, 1024, 512)]
for mode, bands, threads in modes:
    if mode in ("default", "nostream", "slotstreams", "twouploads", "spin0", "spin50"):   # what the product library does: no knob set
        for k in ("SADHIP_TUNE_BANDKB", "SADHIP_TUNE_COPYTHREADS", "SADHIP_TUNE_STREAM", "SADHIP_TUNE_SLOTSTREAMS", "SADHIP_TUNE_PAIRUP"): os.environ.pop(k, None)
        if mode == "twouploads": os.environ["SADHIP_TUNE_PAIRUP"] = "0"
        if mode == "slotstreams": os.environ["SADHIP_TUNE_SLOTSTREAMS"] = "1"
        if mode == "nostream": os.environ["SADHIP_TUNE_STREAM"] = "0"
    else:
        os.environ["SADHIP_TUNE_BANDKB"] = str(max(bands, 4))
        os.environ["SADHIP_TUNE_COPYTHREADS"] = str(threads)
    ctx = capi.Context(0, W, H, 1)
    if mode.startswith("spin"): ctx.set_copy_spin_us(int(mode[4:]))
    o = np.zeros((H, W), np.uint8)
    m1, lo1 = med(lambda: ctx.disparity(ls[0], rs[0], B, D, out=o))
    k, t = ctx.last_timing()
    it = iter(range(10 ** 9))
    def cold():                                               # a different pair and output every call: nothing in the caches
        i = next(it) % NB
        ctx.disparity(ls[i], rs[i], B, D, out=os_[i])
    mc, loc = med(cold, 48)
    if ref is None: ref = o.copy()
    ok = bool(np.array_equal(o, ref))
    ctx.disparity_batch(ls, rs, B, D, outs=os_)
    t0 = time.perf_counter(); ctx.disparity_batch(ls, rs, B, D, outs=os_); dt = time.perf_counter() - t0
    ok = ok and bool(np.array_equal(os_[0], ref))
    m2, lo2 = med(lambda: ctx.disparity_rgba(rgba[0], rgba[1], B, D), 20)
    m3, lo3 = med(lambda: ctx.gray(rgba[0]), 20)
    dev_map = ctx.disparity_rgba(rgba[0], rgba[1], B, D)
    m4, lo4 = med(lambda: ctx.disparity_rgba(rgba[0], rgba[1], B, D, capi.FLAG_HOST_GRAY), 20)
    ok = ok and bool(np.array_equal(ctx.disparity_rgba(rgba[0], rgba[1], B, D, capi.FLAG_HOST_GRAY), dev_map))
    m5, lo5 = med(lambda: ctx.disparity_rgba(rgba_any[0], rgba_any[1], B, D, capi.FLAG_HOST_GRAY), 20)
    ok = ok and bool(np.array_equal(ctx.disparity_rgba(rgba_any[0], rgba_any[1], B, D, capi.FLAG_HOST_GRAY), ctx.disparity_rgba(rgba_any[0], rgba_any[1], B, D)))
    print(f"{mode:7s} band={bands} KiB threads={threads}: single call {m1:.3f} ms (min {lo1:.3f}, kernel {k:.3f}); over 64 different pairs {mc:.3f} ms (min {loc:.3f})  batch {NB / dt:7.0f} pairs/s  "
          f"rgba call {m2:.3f} ms (min {lo2:.3f})  rgba call with HOST_GRAY {m4:.3f} ms (min {lo4:.3f}; arbitrary alpha {m5:.3f}, min {lo5:.3f})  "
          f"gray call {m3:.3f} ms (min {lo3:.3f})  same result={ok}", flush=True)
    ctx.close()
