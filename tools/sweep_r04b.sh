#!/bin/bash
# Round 4 sweep, second part (GPU box): mixed-chunk plans against the uniform plans they replace, 1080p, 64 pairs; and
# the headline configs (SGPR allocation of the two-form kernels moved with the new kernel arguments).
run() { echo -n "$1 | "; shift; env "$@" 2>&1 | grep -E "us/frame|rror" | tail -1; }
B="python tools/bench_cfg.py 1920 1080"
for blk in 16 21 31; do
  for d in 48 80 112 176; do
    run "b$blk D$d default (mixed)  " SADHIP_TUNE_VERBOSE=0 $B $blk $d 64
    run "b$blk D$d no mixed plans   " SADHIP_TUNE_MIX=0 $B $blk $d 64
  done
  run "b$blk D96 default          " SADHIP_TUNE_VERBOSE=0 $B $blk 96 64
  run "b$blk D128 default         " SADHIP_TUNE_VERBOSE=0 $B $blk 128 64
done
python tools/bench_cfg.py 1920 1080 9 64 128
python tools/bench_cfg.py 1920 1080 9 64 1024
python tools/bench_cfg.py 3840 2160 15 128 16
python tools/bench_cfg.py 1920 1080 15 64 64
python tools/bench_cfg.py 1920 1080 16 64 64
