#!/usr/bin/env python3
"""Stress of the round-3 boundary paths on the GPU box:
  (1) sadhip_disparity_lr_u8 -- random left / right sizes (smaller, equal, larger in either axis), parameters and
      regions against the literal restatement run with the right image's own size (oracle.disparity_literal_lr);
  (2) page-locked batches -- random frame counts, column windows inside wider page-locked images (row stride > w),
      scattered / adjacent frame order, tight and windowed outputs against the same batch from pageable copies, with
      sentinel bytes around every output window.
usage: stress_boundary.py [cases] [seed]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import steroscopic_hardware_amd as sha, oracle
from steroscopic_hardware_amd import capi

def run(ncases=200, seed=0):
    """Returns (failing left/right-size cases, failing page-locked layout cases)."""
    rng = np.random.default_rng(seed)
    ctx = capi.Context(0, 1024, 512, 1)
    bad = 0
    for i in range(ncases):                                          # ---- (1) left / right sizes
        lw, lh = int(rng.integers(1, 260)), int(rng.integers(1, 90))
        rw = int(rng.choice([lw, rng.integers(1, lw + 1), rng.integers(lw, lw + 40)]))
        rh = int(rng.choice([lh, rng.integers(1, lh + 1), rng.integers(lh, lh + 20)]))
        block, D = int(rng.integers(1, 20)), int(rng.choice([1, 7, 16, 32, 64, 100]))
        L = rng.integers(0, 256, (lh, lw), dtype=np.uint8)
        R = rng.integers(0, 256, (rh, rw), dtype=np.uint8)
        hh, ww = min(lh, rh), min(lw, rw)
        if rng.random() < 0.5 and ww > 6:                            # a true shift where both exist
            s = int(rng.integers(0, min(D, ww - 1) + 1)); R[:hh, :ww - s] = L[:hh, s:ww]
        reg = None
        if rng.random() < 0.4 and lw > 3 and lh > 3:
            x0 = int(rng.integers(0, lw - 1)); y0 = int(rng.integers(0, lh - 1))
            reg = (x0, y0, int(rng.integers(x0 + 1, lw + 1)), int(rng.integers(y0 + 1, lh + 1)))
        raw = bool(rng.integers(0, 2))
        try:
            got = ctx.disparity_lr(L, R, block, D, region=reg, flags=capi.FLAG_RAW if raw else 0)
            exp = oracle.disparity_literal_lr(L, R, block, D, region=reg, raw=raw)
        except Exception as e:
            bad += 1; print("ERROR lr", i, (lw, lh), (rw, rh), block, D, reg, e, flush=True); continue
        if not np.array_equal(got, exp):
            bad += 1; print("MISMATCH lr", i, (lw, lh), (rw, rh), block, D, reg, raw, int((got != exp).sum()), flush=True)
    print(f"lr: {ncases} cases, {bad} bad", flush=True)

    bad2 = 0
    with capi.HostBuffer(3 * 24 * 200 * 400) as hb:                  # ---- (2) page-locked batch layouts
        for i in range(ncases // 2):
            n = int(rng.integers(1, 25)); W, H = int(rng.integers(60, 401)), int(rng.integers(12, 201))
            w = int(rng.integers(40, W + 1)); x0 = int(rng.integers(0, W - w + 1))
            block, D = int(rng.choice([3, 7, 9, 15, 16])), int(rng.choice([16, 32, 64]))
            blk = hb.view((3, n, H, W))
            blk[0] = rng.integers(0, 256, (n, H, W), dtype=np.uint8); blk[1] = rng.integers(0, 256, (n, H, W), dtype=np.uint8); blk[2] = 0x5A
            order = list(rng.permutation(n)) if rng.random() < 0.5 else list(range(n))
            tight_out = rng.random() < 0.5
            win = slice(x0, x0 + w)
            ls, rs = [blk[0, k][:, win] for k in order], [blk[1, k][:, win] for k in order]
            ref = ctx.disparity_batch([np.ascontiguousarray(a) for a in ls], [np.ascontiguousarray(a) for a in rs], block, D)      # pageable path
            try:
                if tight_out:
                    tight = hb.view((3, n, H, W))[2].reshape(-1)[:n * H * w].reshape(n, H, w)
                    outs = ctx.disparity_batch(ls, rs, block, D, outs=[tight[k] for k in range(n)])
                    ok = all(np.array_equal(outs[k], ref[k]) for k in range(n))
                else:
                    outs = ctx.disparity_batch(ls, rs, block, D, outs=[blk[2, k][:, win] for k in order])
                    ok = all(np.array_equal(outs[k], ref[k]) for k in range(n))
                    ok = ok and all((blk[2, k][:, :x0] == 0x5A).all() and (blk[2, k][:, x0 + w:] == 0x5A).all() for k in range(n))
            except Exception as e:
                bad2 += 1; print("ERROR pinned", i, n, (W, H, w, x0), block, D, e, flush=True); continue
            if not ok:
                bad2 += 1; print("MISMATCH pinned", i, n, (W, H, w, x0), block, D, "tight" if tight_out else "window", flush=True)
    print(f"pinned: {ncases // 2} cases, {bad2} bad", flush=True)
    ctx.close()
    return bad, bad2


if __name__ == "__main__":
    sys.exit(1 if any(run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)) else 0)
