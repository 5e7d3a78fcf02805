#!/bin/bash
cd "$(dirname "$0")/.."
for c in "3840 2160 15 128 16" "3840 2160 15 128 32" "3840 2160 15 128 8" "1920 1080 9 64 8" "1920 1080 9 64 16" "1920 1080 9 64 48" "1920 1080 9 64 128" "1920 1080 15 64 64" "1920 1080 3 16 64" "1280 720 9 64 32"; do
  line="$c :"
  for t in 0 2 3 4 0 3; do
    v=$(SADHIP_TUNE_TAPER=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  taper$t $v"
  done
  echo "$line"
done
SADHIP_TUNE_VERBOSE=1 SADHIP_TUNE_TAPER=3 python tools/bench_cfg.py 3840 2160 15 128 16 2>&1 | grep "sadhip:" | sort | uniq -c
