#!/bin/bash
# Timing survey over the reference UI's parameter grid (1080p, 64 resident pairs per launch): one
# bench_cfg.py line per cell, every cell checked against the oracle on two row bands.
# usage (GPU box): tools/survey_grid.sh > gpurun_out/grid.txt
for b in 3 7 9 11 15 16 21 31; do
  for d in 16 32 48 64 96 128 192 256; do
    timeout -k 10 120 python tools/bench_cfg.py 1920 1080 $b $d 64 2>&1 | grep us/frame
  done
done
