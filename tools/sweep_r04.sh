#!/bin/bash
# Round 4 sweep (GPU box): forced plans of the unpacked kernels on the u16 ring, chunk sizes 4 and 8, 1080p, 64 pairs.
# usage: tools/sweep_r04.sh > gpurun_out/r04_sweep.txt
run() { echo -n "$1 | "; shift; env "$@" 2>&1 | grep -E "us/frame|rror" | tail -1; }
B="python tools/bench_cfg.py 1920 1080"
for blk in 16 21; do
  run "b$blk D16 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 16 64
  run "b$blk D16 nd8 nw4 (round 3)  " SADHIP_TUNE_W16ND4=0 $B $blk 16 64
  run "b$blk D48 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 48 64
  run "b$blk D48 nd8 nw7 no nd4     " SADHIP_TUNE_W16ND4=0 $B $blk 48 64
  run "b$blk D48 nd4 nw12           " SADHIP_TUNE_ND=4 SADHIP_TUNE_NW=12 $B $blk 48 64
  run "b$blk D48 nd8 nw8            " SADHIP_TUNE_ND=8 SADHIP_TUNE_NW=8 $B $blk 48 64
  run "b$blk D48 nd8 nw6            " SADHIP_TUNE_ND=8 SADHIP_TUNE_NW=6 $B $blk 48 64
  run "b$blk D80 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 80 64
  run "b$blk D80 w16 nw<=16        " SADHIP_TUNE_W16NW=16 SADHIP_TUNE_CHUNK=128 $B $blk 80 64
  run "b$blk D80 w16 nd8 nw12      " SADHIP_TUNE_W16NW=16 SADHIP_TUNE_CHUNK=128 SADHIP_TUNE_ND=8 SADHIP_TUNE_NW=12 $B $blk 80 64
  run "b$blk D96 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 96 64
  run "b$blk D96 w16 nw<=16        " SADHIP_TUNE_W16NW=16 SADHIP_TUNE_CHUNK=128 $B $blk 96 64
  run "b$blk D128 default          " SADHIP_TUNE_VERBOSE=0 $B $blk 128 64
  run "b$blk D128 w16 nw<=16       " SADHIP_TUNE_W16NW=16 SADHIP_TUNE_CHUNK=128 $B $blk 128 64
  run "b$blk D64 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 64 64
  run "b$blk D64 nd4 nw16          " SADHIP_TUNE_ND=4 SADHIP_TUNE_NW=16 $B $blk 64 64
  run "b$blk D32 default           " SADHIP_TUNE_VERBOSE=0 $B $blk 32 64
  run "b$blk D32 nd4 nw8           " SADHIP_TUNE_ND=4 SADHIP_TUNE_NW=8 $B $blk 32 64
done
