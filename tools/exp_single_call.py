#!/usr/bin/env python3
"""Round 4: where does one synchronous 1080p call from page-locked memory lose its time on the boxes where it takes 0.44 ms
instead of 0.21 (profiles/r04_bench_line.json)?  Times sadhip_disparity_u8 with inputs / output in page-locked or pageable
memory (four combinations) and the bare copies (hipMemcpy of 2 MB / 4 MB pinned <-> device through torch)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np, torch
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
if os.environ.get("SADHIP_LIB"): capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])
W, H = 1920, 1080
L, R = sha.synth.synth_pair(3, W, H)
ctx = capi.Context(0, W, H, 1)
print("device", capi.device_describe(0))
with capi.HostBuffer(3 * W * H) as hb:
    v = hb.view((3, H, W)); v[0] = L; v[1] = R
    pl, pr, po = v[0], v[1], v[2]
    gl, gr, go = L.copy(), R.copy(), np.zeros((H, W), np.uint8)
    for name, (a, b, o) in (("in pinned, out pinned", (pl, pr, po)), ("in pinned, out pageable", (pl, pr, go)),
                            ("in pageable, out pinned", (gl, gr, po)), ("in pageable, out pageable", (gl, gr, go))):
        for _ in range(3): ctx.disparity(a, b, 9, 64, out=o)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); ctx.disparity(a, b, 9, 64, out=o); ts.append(time.perf_counter() - t0)
        k, t = ctx.last_timing()
        print(f"{name:28s} median {sorted(ts)[10]*1e3:.3f} ms  min {min(ts)*1e3:.3f}  kernel {k:.3f}  lib total {t:.3f}")
# bare copies through torch (pinned tensors)
dev = torch.device("cuda", 0)
for mb in (2, 4):
    hp = torch.empty(mb << 20, dtype=torch.uint8).pin_memory(); dd = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    for direction in ("h2d", "d2h"):
        for _ in range(3):
            (dd.copy_(hp, non_blocking=True) if direction == "h2d" else hp.copy_(dd, non_blocking=True)); torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            (dd.copy_(hp, non_blocking=True) if direction == "h2d" else hp.copy_(dd, non_blocking=True)); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"torch {direction} {mb} MB pinned: median {sorted(ts)[10]*1e6:.0f} us  min {min(ts)*1e6:.0f}")
ctx.close()

# ---- the bench's own sequence, step by step: which step (if any) turns the single call slow? ----------------------
import subprocess
ctx = capi.Context(0, W, H, 1)
def single(tag, a, b, o):
    for _ in range(2): ctx.disparity(a, b, 9, 64, out=o)
    ts = []
    for _ in range(16):
        t0 = time.perf_counter(); ctx.disparity(a, b, 9, 64, out=o); ts.append(time.perf_counter() - t0)
    print(f"{tag:58s} median {sorted(ts)[8]*1e3:.3f} ms  min {min(ts)*1e3:.3f}", flush=True)
nf = 256
with capi.HostBuffer(3 * nf * W * H) as hb:
    blk = hb.view((3, nf, H, W))
    for i in range(nf):
        blk[0, i] = L; blk[1, i] = R
    single("1.6 GB page-locked block, fresh", blk[0, 5], blk[1, 5], blk[2, 5])
    ls, rs, os_ = [blk[0, i] for i in range(nf)], [blk[1, i] for i in range(nf)], [blk[2, i] for i in range(nf)]
    for _ in range(2):
        t0 = time.perf_counter(); ctx.disparity_batch(ls, rs, 9, 64, outs=os_); dt = time.perf_counter() - t0
    print(f"256-frame page-locked batch: {nf / dt:.0f} pairs/s", flush=True)
    single("after the 256-frame batch", blk[0, 7], blk[1, 7], blk[2, 7])
    # the bench's own loop: 32 DIFFERENT frames, one call each -- twice over the same 32 (pages the device has / has not used)
    for rep in range(3):
        ts = []
        for i in range(40, 72):
            t0 = time.perf_counter(); ctx.disparity(blk[0, i], blk[1, i], 9, 64, out=blk[2, i]); ts.append(time.perf_counter() - t0)
        print(f"32 different page-locked frames, pass {rep}: mean {sum(ts)/len(ts)*1e3:.3f} ms  median {sorted(ts)[16]*1e3:.3f}  max {max(ts)*1e3:.3f}", flush=True)
    with capi.HostBuffer(3 * 32 * W * H) as hb2:                       # a block the batch has never touched
        b2 = hb2.view((3, 32, H, W))
        for i in range(32):
            b2[0, i] = L; b2[1, i] = R
        for rep in range(2):
            ts = []
            for i in range(32):
                t0 = time.perf_counter(); ctx.disparity(b2[0, i], b2[1, i], 9, 64, out=b2[2, i]); ts.append(time.perf_counter() - t0)
            print(f"32 frames of a block no device call has touched, pass {rep}: mean {sum(ts)/len(ts)*1e3:.3f} ms  median {sorted(ts)[16]*1e3:.3f}  max {max(ts)*1e3:.3f}", flush=True)
    time.sleep(10)
    single("after 10 s idle", blk[0, 9], blk[1, 9], blk[2, 9])
    prof = "/opt/rocm/bin/rocprofv3"
    r = subprocess.run([prof, "--kernel-trace", "--kernel-include-regex", "k_colwave", "--pmc", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "--output-format", "csv",
                        "-d", "/tmp/exp_pmc", "-o", "pmc", "--", sys.executable, os.path.join(REPO, "bench.py"), "--headline-only", "--steps", "2", "--warmup", "1",
                        "--frames-per-gpu", "256"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=200)
    print("child rocprofv3 --pmc pass exit", r.returncode, flush=True)
    single("after a child process under rocprofv3 --pmc", blk[0, 11], blk[1, 11], blk[2, 11])
    single("pageable, same moment", gl, gr, go)
ctx.close()
