"""steroscopic-hardware_amd -- MI355X (gfx950) implementation of the SAD block-matching disparity
path of conneroisu/steroscopic-hardware (Go package pkg/despair).

Layout (only what the path needs):
  csrc/        hand-written HIP kernels + the C-ABI host code -> lib/libsadhip.so (include/sadhip.h)
  capi.py      ctypes binding of that C ABI (what the tests, bench.py and smoke() call)
  despair.py   Python mirror of the Go API surface (RunSad, SetupConcurrentSAD, ...) over capi
  host/        C++ mirror of the Go API + the cgo shim source a maintainer drops into pkg/despair
  synth.py     the synthetic stereo-pair generator of SURVEY.md section 8(d) (bench/test input)
  shard.py     frame-stream sharding across ranks + the parameter broadcast (multi-GPU)

The product path has no CPU fallback: importing ``capi`` without a built libsadhip.so, or
calling it without a GPU, fails loudly.
"""
from . import capi, despair, shard, synth  # noqa: F401
from .capi import SadHipError, Context, build_library  # noqa: F401

__all__ = ["capi", "despair", "shard", "synth", "SadHipError", "Context", "build_library"]
