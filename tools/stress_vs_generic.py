#!/usr/bin/env python3
"""Stress campaign on the GPU box: the column-wave kernel against the generic kernel (an
independent implementation, itself pinned to the oracle by tests/) on random shapes, parameters,
image statistics, flags and regions.  usage: stress_vs_generic.py [cases] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi

def run(ncases=300, seed=0):
    """Returns the number of failing cases (0 = every case agreed)."""
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0, 4096, 2200, 1)
    bad = 0
    kinds = {}
    t0 = time.time()
    for i in range(ncases):
        w = int(rng.choice([rng.integers(1, 64), rng.integers(64, 600), rng.integers(600, 2100)]))
        h = int(rng.choice([rng.integers(1, 20), rng.integers(20, 120), rng.integers(120, 260)]))
        block = int(rng.integers(1, 34))
        if rng.random() < 0.3:                                      # round 3's new forms: BlockSize 15 (two-form, h = 7), 16 / 17 (suffix-free window sums), 9
            block = int(rng.choice([9, 14, 15, 15, 16, 16, 17]))
            w = int(rng.choice([w, rng.integers(240, 800), rng.integers(800, 4000)]))
        r4 = rng.random() < 0.2                                     # round 4: u16 ring with chunks of 4, mixed-chunk workgroups (BlockSize >= 16, D = 16 mod 32)
        if r4:
            block = int(rng.integers(16, 32)) if rng.random() < 0.5 else int(rng.choice([3, 5, 7, 9, 9, 11, 13, 15, 15]))   # ... and the packed (ND, ND - 4) plans
            w = int(rng.choice([rng.integers(300, 700), rng.integers(700, 2100)]))
        D = int(rng.choice([rng.integers(1, 20), rng.choice([16, 32, 48, 64, 96, 128, 160, 192, 256]), rng.integers(20, 400)]))
        if r4:
            D = int(rng.choice([16, 48, 80, 112, 144, 176, 208, 240, 47, 79]))
        style = int(rng.integers(0, 6))
        if style == 0:
            L = rng.integers(0, 256, (h, w), dtype=np.uint8); R = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif style == 1:                                            # shifted copy + noise: realistic minima
            L = rng.integers(0, 256, (h, w), dtype=np.uint8); s = int(rng.integers(0, max(1, min(D, w))))
            R = np.roll(L, -s, axis=1); R = (R ^ (rng.integers(0, 16, (h, w)) == 0).astype(np.uint8)).astype(np.uint8)
        elif style == 2:                                            # flat: ties everywhere
            L = np.full((h, w), int(rng.integers(0, 256)), np.uint8); R = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        elif style == 3:                                            # extremes: largest SADs, bands
            L = np.full((h, w), 255, np.uint8); R = np.zeros((h, w), np.uint8); R[:, rng.integers(0, w):] = 255
        elif style == 4:                                            # smooth gradients: many near ties
            x = np.arange(w)[None, :]; y = np.arange(h)[:, None]
            L = ((x * 3 + y * 5) % 256).astype(np.uint8); R = ((x * 3 + y * 5 + int(rng.integers(0, 40))) % 256).astype(np.uint8)
        else:                                                       # low-amplitude texture
            L = rng.integers(100, 104, (h, w), dtype=np.uint8); R = rng.integers(100, 104, (h, w), dtype=np.uint8)
        flags = capi.FLAG_RAW if (rng.random() < 0.5 and D <= 255) else 0
        kind = capi.kernel_kind(w, h, block, D)
        kinds[kind] = kinds.get(kind, 0) + 1
        try:
            if rng.random() < 0.25 and w > 4 and h > 4:
                x0 = int(rng.integers(0, w - 1)); x1 = int(rng.integers(x0 + 1, w + 1)); y0 = int(rng.integers(0, h - 1)); y1 = int(rng.integers(y0 + 1, h + 1))
                a = ctx.disparity_region(L, R, block, D, (x0, y0, x1, y1), flags)
                b = ctx.disparity_region(L, R, block, D, (x0, y0, x1, y1), flags | capi.FLAG_FORCE_GENERIC)
            else:
                a = ctx.disparity(L, R, block, D, flags); b = ctx.disparity(L, R, block, D, flags | capi.FLAG_FORCE_GENERIC)
        except Exception as e:
            bad += 1; print("ERROR", i, w, h, block, D, style, flags, kind, e, flush=True); continue
        if not np.array_equal(a, b):
            bad += 1
            ys, xs = np.nonzero(a != b)
            print("MISMATCH", i, (w, h, block, D), "style", style, "flags", flags, "kind", kind, len(xs), "px; first", (int(xs[0]), int(ys[0])), int(a[ys[0], xs[0]]), int(b[ys[0], xs[0]]), flush=True)
        if i % 50 == 49:
            print(f"{i+1} cases, {bad} bad, {time.time()-t0:.0f}s", flush=True)
    print("done:", ncases, "cases,", bad, "bad; kinds", kinds)
    ctx.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
