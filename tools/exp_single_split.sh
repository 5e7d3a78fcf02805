#!/bin/bash
# Round 5, the judge's item 5: what a candidate split over two workgroup sets could reach for ONE 1080p pair, WITHOUT building it.
# A split launch (65 candidates as 32 + 33 on twice the workgroups, merged through a key scratch) cannot be faster than a launch of
# the same workgroups that does no merging at all: a TWO-pair batch at MaxDisparity 31 (4 waves x 8 candidates, the half's plan)
# with the split launch's strip count.  Strip counts 45 / 60 / 80 / 90 = 720 / 960 / 1280 / 1440 workgroups (8 tiles x strips x 2).
# Next to it: the single pair as it runs today (strip sweep), 8 waves x 8 candidates per workgroup, and the same for 16 / 64.
# usage (GPU box): tools/exp_single_split.sh > gpurun_out/r05_exp_single_split.txt
cd "$(dirname "$0")/.."
run() { timeout -k 10 100 python tools/bench_cfg.py "$@" 2>/dev/null | grep us/frame; }
echo "== today: single pair 9/64 (and the default plan's strips)"; run 1920 1080 9 64 1
for s in 30 45 60 72 90 108 135; do echo -n "strips=$s  "; SADHIP_TUNE_STRIPS=$s run 1920 1080 9 64 1; done
echo "== bound of the split launch: two pairs at 9/31 (one half's plan), the split launch of ONE real pair takes at least 2 x the us/frame printed here (both halves are in the launch)"
for s in 36 45 60 80 90; do echo -n "strips=$s  "; SADHIP_TUNE_STRIPS=$s run 1920 1080 9 31 2; done
echo "== 8 waves x 8 candidates in one workgroup (no merge needed), single pair 9/64"
for s in 32 48 64 90; do echo -n "strips=$s  "; SADHIP_TUNE_ND=8 SADHIP_TUNE_NW=8 SADHIP_TUNE_STRIPS=$s run 1920 1080 9 64 1; done
echo "== defaults 16/64: today, and the split bound (two pairs at 16/31)"
run 1920 1080 16 64 1
for s in 30 45 60 90; do echo -n "strips=$s  "; SADHIP_TUNE_STRIPS=$s run 1920 1080 16 64 1; done
for s in 24 32 45 60; do echo -n "split-bound strips=$s  "; SADHIP_TUNE_STRIPS=$s run 1920 1080 16 31 2; done
echo "== config 5, 16 pairs: strips"
for s in 4 6 8 9 12; do echo -n "strips=$s  "; SADHIP_TUNE_STRIPS=$s run 3840 2160 15 128 16; done
