#!/bin/bash
# Round 5: the issue priority's quarter boundaries dithered (default, SADHIP_TUNE_PRIO=1) against plain quarters (=2) and oldest-first only (=0).
cd "$(dirname "$0")/../.."
for c in "1920 1080 9 64 1" "1920 1080 9 64 2" "1920 1080 9 64 4" "1920 1080 9 64 16" "1920 1080 9 64 48" "1920 1080 9 64 128" "1920 1080 9 64 1024" "1920 1080 16 64 1" "1920 1080 16 64 16" "1920 1080 16 64 64" \
         "3840 2160 15 128 1" "3840 2160 15 128 16" "3840 2160 15 128 32" "1920 1080 15 64 64" "1920 1080 31 64 16" "640 480 7 32 128" "640 480 16 64 1" "640 480 7 32 1" "1280 720 9 64 8" "1920 1080 3 16 64" "1920 1080 16 128 64"; do
  line="$c :"
  for t in 0 2 1 2 1; do
    v=$(SADHIP_TUNE_PRIO=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  prio=$t $v"
  done
  echo "$line"
done
