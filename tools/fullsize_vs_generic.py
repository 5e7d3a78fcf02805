import os, sys, time
sys.path[:0] = ["/root/repo"]
import numpy as np
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
ctx = capi.Context(0, 3840, 2160, 1)
for (seed, w, h, block, D) in [(3, 1920, 1080, 9, 64), (5, 3840, 2160, 15, 128), (3, 1920, 1080, 16, 64), (3, 1920, 1080, 31, 256), (3, 1920, 1080, 11, 256), (3, 1920, 1080, 13, 96), (2, 640, 480, 7, 32), (5, 3840, 2160, 9, 64), (5, 3840, 2160, 25, 64)]:
    L, R = sha.synth.synth_pair(seed, w, h)
    t0 = time.time(); a = ctx.disparity(L, R, block, D); t1 = time.time(); b = ctx.disparity(L, R, block, D, capi.FLAG_FORCE_GENERIC); t2 = time.time()
    print(w, h, block, D, "equal", bool(np.array_equal(a, b)), "fast %.1f ms generic %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "kind", capi.kernel_kind(w, h, block, D), flush=True)
