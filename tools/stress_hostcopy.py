#!/usr/bin/env python3
"""Stress of the staged host path (round 4: caller memory that is not page-locked is copied into / out of page-locked planes
of the context by its copy threads, in bands above 8 MB) on the GPU box.  Random frame sizes around the thresholds (the copy
threads start at 512 KB per copy, bands at 12 MB per plane), row strides, base alignments, regions with column offsets,
batch lengths, decoded pictures with 3 / 4 channels and a row stride (device conversion and, round 5, SADHIP_FLAG_HOST_GRAY: every
alpha class of the host conversion), a guard page right behind the last input byte --
against the device-resident entry point on the same bytes (sadhip_disparity_dev_u8, verified by the parity tests), with
sentinel bytes around every output window.
usage: stress_hostcopy.py [cases] [seed]"""
import ctypes, mmap, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle"), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
import torch
import steroscopic_hardware_amd as sha, oracle
from steroscopic_hardware_amd import capi
from _pinned import to_device, to_host


def _guarded(a):
    """copy of a 2-D u8 array whose last byte is the last byte before a PROT_NONE page"""
    page = mmap.PAGESIZE
    n = a.size
    body = (n + page - 1) // page * page
    m = mmap.mmap(-1, body + page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    assert ctypes.CDLL(None, use_errno=True).mprotect(ctypes.c_void_p(base + body), ctypes.c_size_t(page), 0) == 0
    g = np.frombuffer(m, np.uint8, n, body - n).reshape(a.shape)
    g[...] = a
    return g, m


def run(ncases=40, seed=0):
    rng = np.random.default_rng(seed)
    MW, MH = 6200, 2300
    ctx = capi.Context(0, MW, MH, 1)
    bad = 0
    for i in range(ncases):
        kind = rng.choice(["frame", "frame", "region", "batch", "pixels"])
        w = int(rng.choice([rng.integers(200, 900), rng.integers(900, 3000), rng.integers(3000, MW + 1)]))
        h = int(rng.choice([rng.integers(64, 400), rng.integers(400, MH + 1)]))
        if kind == "pixels": w, h = min(w, 3900), min(h, 2200)
        block, D = int(rng.choice([3, 5, 8, 9])), int(rng.choice([8, 16, 32]))
        padl, padr, pado = (int(rng.choice([0, 1, 13, 64, 300])) for _ in range(3))
        offl, offr, offo = (int(rng.integers(0, p + 1)) for p in (padl, padr, pado))
        bigl = rng.integers(0, 256, (h, w + padl), dtype=np.uint8)
        bigr = rng.integers(0, 256, (h, w + padr), dtype=np.uint8)
        L, R = bigl[:, offl:offl + w], bigr[:, offr:offr + w]
        s = int(rng.integers(0, D + 1))
        if w > s + 4: R[:, :w - s] = L[:, s:]                         # a true shift
        Lc, Rc = np.ascontiguousarray(L), np.ascontiguousarray(R)
        dL, dR = to_device(Lc[None]), to_device(Rc[None])
        dO = torch.empty_like(dL)
        ctx.disparity_dev(dL, dR, dO, block, D)
        torch.cuda.synchronize()
        exp = to_host(dO[0])
        ok = True
        if kind == "frame":
            out = np.full((h, w + pado), 0xA5, np.uint8)
            win = out[:, offo:offo + w]
            if rng.random() < 0.4:
                gl, ml = _guarded(Lc); gr, mr = _guarded(Rc)
                ctx.disparity(gl, gr, block, D, out=win)
                del gl, gr; ml.close(); mr.close()
            else:
                ctx.disparity(L, R, block, D, out=win)
            ok = np.array_equal(win, exp) and (out[:, :offo] == 0xA5).all() and (out[:, offo + w:] == 0xA5).all()
        elif kind == "region":
            for _ in range(3):
                x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
                x1, y1 = int(rng.integers(x0 + 1, w + 1)), int(rng.integers(y0 + 1, h + 1))
                ok = ok and np.array_equal(ctx.disparity_region(L, R, block, D, (x0, y0, x1, y1)), exp[y0:y1, x0:x1])
        elif kind == "batch":
            n = int(rng.integers(1, 8))
            wide = np.full((n, h, w + pado), 0x5A, np.uint8)
            ctx.disparity_batch([L] * n, [R] * n, block, D, outs=[wide[k, :, offo:offo + w] for k in range(n)])
            ok = all(np.array_equal(wide[k, :, offo:offo + w], exp) for k in range(n)) and \
                (wide[:, :, :offo] == 0x5A).all() and (wide[:, :, offo + w:] == 0x5A).all()
        else:
            ch = int(rng.choice([3, 4]))
            pic = rng.integers(0, 256, (2, h, w + padl, ch), dtype=np.uint8)
            if ch == 4:                                               # alpha classes of the host conversion: opaque, near-opaque, sparse, arbitrary
                amode = int(rng.integers(0, 4))
                if amode == 0: pic[..., 3] = 255
                elif amode == 1: pic[..., 3] = rng.integers(251, 256, pic.shape[:3])
                elif amode == 2: pic[..., 3] = np.where(rng.integers(0, 9, pic.shape[:3]) == 0, pic[..., 3], 255)
            lp, rp = pic[0, :, offl:offl + w], pic[1, :, offl:offl + w]
            gl_, gr_ = ctx.gray(lp), ctx.gray(rp)
            want = (oracle.gray_from_nrgba if ch == 4 else oracle.gray_from_rgb8)(np.ascontiguousarray(lp))
            dL, dR = to_device(gl_[None]), to_device(gr_[None])
            ctx.disparity_dev(dL, dR, dO, block, D)
            torch.cuda.synchronize()
            ok = np.array_equal(gl_, want) and np.array_equal(ctx.disparity_rgba(lp, rp, block, D), to_host(dO[0]))
            # round 5: the same pictures converted by the copy threads while they stage (SADHIP_FLAG_HOST_GRAY)
            ok = ok and np.array_equal(ctx.disparity_rgba(lp, rp, block, D, capi.FLAG_HOST_GRAY), to_host(dO[0]))
        if not ok:
            bad += 1
            print(f"MISMATCH case {i}: {kind} {w}x{h} b{block} D{D} pads {padl},{padr},{pado} offs {offl},{offr},{offo}", flush=True)
    ctx.close()
    print(f"stress_hostcopy: {ncases} cases, {bad} mismatches", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
