import sys, time
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import steroscopic_hardware_amd as sha
d = sha.despair
L, R = sha.synth.synth_pair(31, 640, 480)
d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=32))
inp, outp = d.SetupConcurrentSAD(32)
def frame():
    n = 0
    for y in range(0, 480, 3):
        inp.send(d.InputChunk(L, R, (0, y, 640, min(y + 3, 480)))); n += 1
    return d.AssembleDisparityMap(outp, (0, 0, 640, 480), n)
frame(); frame()
ts = []
for _ in range(10):
    t0 = time.perf_counter(); got = frame(); ts.append(time.perf_counter() - t0)
print(f"channel API, 640x480, 160 chunks per frame (examples/run.stream.go's shape), python twin: median {sorted(ts)[5]*1e3:.2f} ms per frame")
inp.close()
# what one chunk costs as a device call of its own (the worker before it gathered chunks)
with sha.despair._context(640, 480) as c:
    for _ in range(5): c.disparity_region(L, R, 16, 32, (0, 100, 640, 103))
    t0 = time.perf_counter()
    for y in range(0, 480, 3): c.disparity_region(L, R, 16, 32, (0, y, 640, min(y + 3, 480)))
    dt = time.perf_counter() - t0
print(f"the same 160 chunks as 160 device calls: {dt*1e3:.2f} ms per frame ({dt/160*1e6:.0f} us per call)")
t0 = time.perf_counter(); d.RunSadFrame(L, R, d.Parameters(16, 32)); print(f"one frame-level call: {(time.perf_counter()-t0)*1e3:.2f} ms")
