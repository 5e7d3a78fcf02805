#!/usr/bin/env python3
"""Stress of the device-resident batch entry point: random batch sizes, row / frame strides
(views into larger tensors), parameters; column-wave kernel vs generic kernel through the same call.
usage: stress_dev_batch.py [cases] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np, torch
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi

def run(ncases=200, seed=0):
    """Returns the number of failing cases."""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, 2048, 1200, 1)
    bad = 0
    for i in range(ncases):
        n = int(rng.integers(1, 9)); w = int(rng.integers(40, 1400)); h = int(rng.integers(8, 200))
        block = int(rng.integers(2, 32)); D = int(rng.choice([16, 32, 64, 65, 100, 128, 200, 256]))
        if i % 3 == 1:                                              # the BlockSize-9 tile split (>= 48 pairs): two kernels per pass
            block = int(rng.choice([8, 9, 14, 15, 15, 16, 17])); n = int(rng.choice([3, 24, 48, 49, 64, 77])); w = int(rng.integers(250, 1400)); h = int(rng.integers(8, 40))
        padw = int(rng.integers(0, 70)); padh = int(rng.integers(0, 9))
        big_l = torch.randint(0, 256, (n, h + padh, w + padw), dtype=torch.uint8, device=dev)
        big_r = torch.randint(0, 256, (n, h + padh, w + padw), dtype=torch.uint8, device=dev)
        if i % 3 == 0:
            big_r[:, :, : w + padw - 5] = big_l[:, :, 5:]          # a true shift of 5
        off = int(rng.integers(0, padw + 1))
        L, R = big_l[:, :h, off:off + w], big_r[:, :h, off:off + w]        # strided views: row stride w+padw, frame stride (h+padh)*(w+padw)
        out_a = torch.zeros((n, h + 1, w + 3), dtype=torch.uint8, device=dev)[:, :h, :w]
        out_b = torch.zeros_like(out_a)
        flags = capi.FLAG_RAW if (D <= 255 and rng.random() < 0.5) else 0
        try:
            ctx.disparity_dev(L, R, out_a, block, D, flags)
            ctx.disparity_dev(L, R, out_b, block, D, flags | capi.FLAG_FORCE_GENERIC)
            torch.cuda.synchronize()
        except Exception as e:
            bad += 1; print("ERROR", i, n, w, h, block, D, e, flush=True); continue
        if not torch.equal(out_a, out_b):
            bad += 1; print("MISMATCH", i, (n, w, h, block, D), int((out_a != out_b).sum()), flush=True)
    print("done:", ncases, "cases,", bad, "bad")
    ctx.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
