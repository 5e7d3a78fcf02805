#!/usr/bin/env python3
"""Runs the plan-shape grid of tests/test_parity_gpu.py::test_every_plan_shape and prints every mismatch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import steroscopic_hardware_amd as sha, oracle
from steroscopic_hardware_amd import capi
ctx = capi.Context(0, 3840, 2160, 1)
rng = np.random.default_rng(7)
w = 420
blocks = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else (3, 5, 7, 9, 11, 13, 15, 16, 17)
for block in blocks:
    hh = block // 2; hgt = 2 * hh + 6
    Lr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8); Rr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
    Lx = np.full((hgt, w), 255, np.uint8); Rx = np.zeros((hgt, w), np.uint8)
    Rx[:, w // 3:] = 255; Rx[:, 2 * w // 3:] = 0
    for D in (16, 17, 18, 19, 20, 33, 47, 64, 65, 70, 100, 128, 130, 192, 255, 256):
        for name, (L, R) in (("random", (Lr, Rr)), ("extreme", (Lx, Rx))):
            raw = D <= 255
            exp = oracle.disparity_closed(L, R, block, D, raw=raw)
            got = ctx.disparity(L, R, block, D, capi.FLAG_RAW if raw else 0)
            if not np.array_equal(got, exp):
                ys, xs = np.nonzero(got != exp)
                print(f"MISMATCH block={block} D={D} {name} kind={capi.kernel_kind(w,hgt,block,D)}: {len(xs)} px; x range {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}; first (x={xs[0]},y={ys[0]}) got {got[ys[0],xs[0]]} exp {exp[ys[0],xs[0]]}; xs sample {sorted(set(xs.tolist()))[:12]}")
print("done")
