#!/bin/bash
# Round 4, config 5 (3840x2160 B15 D128, 16 pairs) and the defaults (1080p B16 D64, 64 pairs): what the 129th / 65th
# candidate, the strip count and a two-pass split cost.  usage: tools/sweep_r04c.sh > gpurun_out/r04_sweep_c.txt
run() { echo -n "$1 | "; shift; env "$@" 2>&1 | grep -E "us/frame|rror" | tail -1; }
C5="python tools/bench_cfg.py 3840 2160 15"
run "c5 D128 default                  " SADHIP_TUNE_VERBOSE=0 $C5 128 16
run "c5 D127 (128 candidates, no odd one)" SADHIP_TUNE_VERBOSE=0 $C5 127 16
run "c5 D128 strips 3                 " SADHIP_TUNE_STRIPS=3 $C5 128 16
run "c5 D128 strips 4                 " SADHIP_TUNE_STRIPS=4 $C5 128 16
run "c5 D128 strips 5                 " SADHIP_TUNE_STRIPS=5 $C5 128 16
run "c5 D128 strips 8                 " SADHIP_TUNE_STRIPS=8 $C5 128 16
run "c5 D128 strips 9                 " SADHIP_TUNE_STRIPS=9 $C5 128 16
run "c5 D128 strips 12                " SADHIP_TUNE_STRIPS=12 $C5 128 16
run "c5 D128 two passes of 4 x 16     " SADHIP_TUNE_CHUNK=64 SADHIP_TUNE_NW=4 SADHIP_TUNE_ND=16 $C5 128 16
run "c5 D128 plain form               " SADHIP_TUNE_TRMODE=0 $C5 128 16
run "c5 D128 32 pairs                 " SADHIP_TUNE_VERBOSE=0 $C5 128 32
B16="python tools/bench_cfg.py 1920 1080 16"
run "b16 D64 default                  " SADHIP_TUNE_VERBOSE=0 $B16 64 64
run "b16 D63 (64 candidates)          " SADHIP_TUNE_VERBOSE=0 $B16 63 64
run "b16 D64 strips 2                 " SADHIP_TUNE_STRIPS=2 $B16 64 64
run "b16 D64 strips 3                 " SADHIP_TUNE_STRIPS=3 $B16 64 64
run "b16 D64 strips 4                 " SADHIP_TUNE_STRIPS=4 $B16 64 64
run "b16 D64 strips 6                 " SADHIP_TUNE_STRIPS=6 $B16 64 64
run "b16 D64 1024 pairs               " SADHIP_TUNE_VERBOSE=0 $B16 64 1024
C3="python tools/bench_cfg.py 1920 1080 9"
run "c3 D64 128 pairs                 " SADHIP_TUNE_VERBOSE=0 $C3 64 128
run "c3 D63 128 pairs (64 candidates) " SADHIP_TUNE_VERBOSE=0 $C3 63 128
