#!/bin/bash
# Round 4: weight of a warm-up row in the strip cost model (SADHIP_TUNE_WARMPCT) after the warm-up rows got cheaper.
for c in "1920 1080 9 64 1" "1920 1080 9 64 2" "1920 1080 9 64 4" "1920 1080 9 64 16" "1920 1080 9 64 48" "1920 1080 9 64 128" "1920 1080 16 64 1" "1920 1080 16 64 8" "1920 1080 16 64 64" "3840 2160 15 128 1" "3840 2160 15 128 4" "3840 2160 15 128 16" "640 480 7 32 1" "640 480 7 32 16" "1280 720 9 64 1" "1920 1080 31 64 1" "1920 1080 31 64 16" "1920 1080 3 16 1"; do
  line="$c :"
  for t in 100 70 50 35; do
    v=$(SADHIP_TUNE_WARMPCT=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  warm$t $v"
  done
  echo "$line"
done
