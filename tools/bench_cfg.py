#!/usr/bin/env python3
"""Times one configuration (device-resident batch) and checks a band against the oracle.
usage: bench_cfg.py W H BLOCK MAXDISP [NFRAMES]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _pinned import to_host
import steroscopic_hardware_amd as sha, oracle
if os.environ.get("SADHIP_LIB"): sha.capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])   # other builds
elif any(k.startswith("SADHIP_TUNE_") for k in os.environ):      # the knobs exist only in the diagnostics build (make -C csrc tune)
    sha.capi.LIB_PATH = os.path.join(os.path.dirname(sha.capi.LIB_PATH), "libsadhip_tune.so")
W, H, B, D = (int(v) for v in sys.argv[1:5]); N = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda", 0)
_parts = [sha.synth.synth_batch_torch([1000 + i for i in range(c, min(c + 128, N))], W, H, dev) for c in range(0, N, 128)]   # in chunks of 128: bounded temporaries
Ls, Rs = torch.cat([p[0] for p in _parts]), torch.cat([p[1] for p in _parts]); del _parts
out = torch.empty_like(Ls)
ctx = sha.capi.Context(0, W, H, 1)
for _ in range(2): ctx.disparity_dev(Ls, Rs, out, B, D)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); K = 5
for _ in range(K): ctx.disparity_dev(Ls, Rs, out, B, D)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K / N
L0, R0 = to_host(Ls[0]), to_host(Rs[0])
y0 = H // 2
ok = np.array_equal(to_host(out[0, y0:y0 + 4]), oracle.disparity_closed(L0, R0, B, D, region=(0, y0, W, y0 + 4)))
ok2 = np.array_equal(to_host(out[0, :3]), oracle.disparity_closed(L0, R0, B, D, region=(0, 0, W, 3)))
c = sha.synth.candidates_per_pair(W, H, B, D)
print(f"{W}x{H} b{B} D{D} n={N} kind={sha.capi.kernel_kind(W,H,B,D)}: {ms*1e3:.1f} us/frame  {c/ms/1e6:.0f} Gcand/s  {3*W*H/ms/1e6:.1f} GB/s  parity={ok and ok2}")
