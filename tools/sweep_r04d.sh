#!/bin/bash
# Round 4 sweep, packed mixed-chunk plans (4 waves of ND + 4 of ND - 4) against the uniform plans, 1080p, 64 pairs.
run() { echo -n "$1 | "; shift; env "$@" 2>&1 | grep -E "us/frame|rror" | tail -1; }
B="python tools/bench_cfg.py 1920 1080"
for blk in 3 9 11 13 15; do
  for d in 80 112 144 176 240; do
    run "b$blk D$d mixed  " SADHIP_TUNE_VERBOSE=0 $B $blk $d 64
    run "b$blk D$d uniform" SADHIP_TUNE_MIX=0 $B $blk $d 64
  done
done
run "b16 D48 mixed  " SADHIP_TUNE_VERBOSE=0 $B 16 48 64
run "b9 D64 128     " SADHIP_TUNE_VERBOSE=0 $B 9 64 128
