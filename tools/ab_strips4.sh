#!/bin/bash
# Round 4 A/B of the strip cost model's several-rounds term: the round-2 form (mean of ideal and busiest-CU round counts
# + 0.15) against ideal rounds + a tail of 35 / 50 / 70 % of a workgroup's duration.  usage: tools/ab_strips4.sh > gpurun_out/r04_ab_strips.txt
for c in "1920 1080 9 64 1024" "1920 1080 9 64 128" "1920 1080 9 64 48" "1920 1080 9 64 16" "1920 1080 9 64 4" "3840 2160 15 128 16" "3840 2160 15 128 32" "3840 2160 15 128 4" "1920 1080 16 64 64" "1920 1080 16 64 8" "1920 1080 16 16 64" "640 480 7 32 128" "1920 1080 15 64 64" "1920 1080 31 64 16" "1920 1080 21 96 64" "1920 1080 3 16 64" "1280 720 9 64 32"; do
  line="$c :"
  for t in 0 35 50 70; do
    v=$(SADHIP_TUNE_TAILPCT=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  tail$t $v"
  done
  echo "$line"
done
