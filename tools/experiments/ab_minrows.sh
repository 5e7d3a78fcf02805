#!/bin/bash
cd "$(dirname "$0")/../.."
for c in "640 480 16 64 1" "640 480 7 32 1" "640 480 9 64 1" "1280 720 9 64 1" "1280 720 16 64 1" "1920 1080 16 64 1" "1920 1080 9 64 1" "640 480 16 64 2" "640 480 16 64 4" "320 240 16 64 1" "640 480 31 64 1" "640 480 15 128 1" "960 540 16 64 1" "3840 2160 15 128 1"; do
  line="$c :"
  for t in 8 0 8 0; do
    v=$(SADHIP_TUNE_MINROWS=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  minrows=$t $v"
  done
  echo "$line"
done
