#!/usr/bin/env python3
"""Host-side cost of ONE device-resident call (sadhip_disparity_dev_u8), copies excluded: planning, launch
geometry, kernel launch.  VERDICT r2 item 7b.
  enqueue_us      host time of the call itself (asynchronous: returns after enqueueing), mean over back-to-back calls
  wall_us         call + wait for the result on an idle GPU (what a synchronous caller sees without copies)
  kernel_us       device time of the kernel (HIP events on the launch stream)
  overhead_us     wall - kernel
usage: call_overhead.py [W H BLOCK MAXDISP]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
import steroscopic_hardware_amd as sha
if os.environ.get("SADHIP_LIB"): sha.capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])
W, H, B, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (1920, 1080, 9, 64)
dev = torch.device("cuda", 0)
L, R = sha.synth.synth_batch_torch([3], W, H, dev)
out = torch.empty_like(L)
ctx = sha.capi.Context(0, W, H, 1)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(20): ctx.disparity_dev(L, R, out, B, D, 0, st)
torch.cuda.synchronize()
# host time per enqueue (the GPU lags behind: 200 calls of ~50 us queue up)
t0 = time.perf_counter()
for _ in range(200): ctx.disparity_dev(L, R, out, B, D, 0, st)
enq = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
wall, kern = [], []
for _ in range(100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(); ctx.disparity_dev(L, R, out, B, D, 0, st); e1.record()
    torch.cuda.synchronize()
    wall.append(time.perf_counter() - t0); kern.append(e0.elapsed_time(e1) * 1e-3)
med = lambda v: sorted(v)[len(v) // 2]
# the raw ctypes call without the Python wrapper's tensor checks
l = sha.capi.lib(); h = ctx._h
args = (h, 1, L.data_ptr(), R.data_ptr(), L.stride(0), R.stride(0), W, H, W, W, B, D, 0, out.data_ptr(), out.stride(0), W, st)
t0 = time.perf_counter()
for _ in range(200): l.sadhip_disparity_dev_u8(*args)
raw = (time.perf_counter() - t0) / 200
torch.cuda.synchronize()
print(f"{W}x{H} b{B} D{D}: enqueue {enq*1e6:.1f} us (raw ctypes call {raw*1e6:.1f}), wall {med(wall)*1e6:.1f} us, kernel {med(kern)*1e6:.1f} us, "
      f"overhead (wall - kernel) {(med(wall)-med(kern))*1e6:.1f} us")
