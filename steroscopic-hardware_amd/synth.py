"""Synthetic stereo-pair generator of SURVEY.md section 8(d) (bench / test input only).

splitmix64 stream: draw n (n >= 1) = mix(seed + n * 0x9E3779B97F4A7C15).
  pass 1, row-major, one draw per pixel:   L[y][x] = draw & 0xFF
  pass 2, row-major, two draws a,b per pixel:
      s = 3 + (x*7)//W + (y*5)//H                      (true disparity, 3..13)
      v = L[y][x+s] if x+s < W else a & 0xFF
      if ((b >> 8) & 15) == 0: v ^= (b & 3)
      R[y][x] = v
Seeds: config 2 seed=2 (640x480), config 3 seed=3 (1920x1080), config 4 frame i seed=1000+i,
config 5 seed=5 (3840x2160).  The draw index is a closed form, so numpy and torch can vectorise
it; the tests compare against a sequential C restatement kept with the test infrastructure.
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix_np(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_pair(seed, w, h):
    """numpy version -> (L, R) uint8 [h, w]."""
    n = w * h
    with np.errstate(over="ignore"):
        idx = np.arange(1, 3 * n + 1, dtype=np.uint64)
        z = _mix_np(np.uint64(seed) + idx * _G)
    L = (z[:n] & np.uint64(0xFF)).astype(np.uint8).reshape(h, w)
    a = z[n::2].reshape(h, w)
    b = z[n + 1::2].reshape(h, w)
    xs = np.arange(w, dtype=np.int64)[None, :]
    ys = np.arange(h, dtype=np.int64)[:, None]
    s = 3 + (xs * 7) // w + (ys * 5) // h
    src = xs + s
    inb = src < w
    v = np.where(inb, np.take_along_axis(L, np.minimum(src, w - 1), axis=1), (a & np.uint64(0xFF)).astype(np.uint8))
    flip = ((b >> np.uint64(8)) & np.uint64(15)) == 0
    v = np.where(flip, v ^ (b & np.uint64(3)).astype(np.uint8), v).astype(np.uint8)
    return L, v


def synth_batch_torch(seeds, w, h, device):
    """torch version, generated directly in HBM -> (L, R) uint8 [n, h, w] on ``device``.
    int64 arithmetic wraps like uint64; logical right shifts are emulated by masking."""
    import torch
    n = w * h
    G = 0x9E3779B97F4A7C15 - (1 << 64)
    M1 = 0xBF58476D1CE4E5B9 - (1 << 64)
    M2 = 0x94D049BB133111EB - (1 << 64)

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix(z):
        z = (z ^ lsr(z, 30)) * M1
        z = (z ^ lsr(z, 27)) * M2
        return z ^ lsr(z, 31)

    Ls, Rs = [], []
    idx = torch.arange(1, 3 * n + 1, dtype=torch.int64, device=device)
    xs = torch.arange(w, dtype=torch.int64, device=device)[None, :]
    ys = torch.arange(h, dtype=torch.int64, device=device)[:, None]
    s = 3 + (xs * 7) // w + (ys * 5) // h
    src = xs + s
    inb = src < w
    srcc = torch.clamp(src, max=w - 1)
    for seed in seeds:
        sd = seed if seed < (1 << 63) else seed - (1 << 64)
        z = mix(idx * G + sd)
        L = (z[:n] & 0xFF).to(torch.uint8).reshape(h, w)
        a = z[n::2].reshape(h, w)
        b = z[n + 1::2].reshape(h, w)
        v = torch.where(inb, torch.gather(L, 1, srcc), (a & 0xFF).to(torch.uint8))
        flip = ((b >> 8) & 15) == 0
        v = torch.where(flip, v ^ (b & 3).to(torch.uint8), v)
        Ls.append(L)
        Rs.append(v)
    return torch.stack(Ls), torch.stack(Rs)


def candidates_per_pair(w, h, block, maxdisp):
    """Algorithmic candidate evaluations per pair:
    C = H * sum_x (min(D, max(0, x-hs)) + 1)."""
    hs = block // 2
    xs = np.arange(w)
    return int(h * (np.minimum(maxdisp, np.maximum(0, xs - hs)) + 1).sum())
