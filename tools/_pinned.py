"""Host <-> device copies for the tools, through torch's page-locked allocator only (no device copy
from / into pageable memory)."""
import numpy as np
import torch


def to_device(a, dev="cuda:0"):
    t = torch.from_numpy(np.ascontiguousarray(a))
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    d = p.to(dev)
    torch.cuda.synchronize()
    return d


def to_host(t):
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    torch.cuda.synchronize()
    return p.numpy().copy()
