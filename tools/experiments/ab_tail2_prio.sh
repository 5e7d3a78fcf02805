#!/bin/bash
# with the issue priority by progress on (default): is twice the model's strip count still right for the two-per-CU kernels?
cd "$(dirname "$0")/../.."
for c in "3840 2160 15 128 4" "3840 2160 15 128 8" "3840 2160 15 128 16" "3840 2160 15 128 32" "1920 1080 15 128 16" "1920 1080 13 128 32" "1920 1080 16 64 4" "1920 1080 16 64 16" "1920 1080 16 64 64" "1920 1080 16 64 256" "1920 1080 9 128 64"; do
  line="$c :"
  for t in 0 1 0 1; do
    v=$(SADHIP_TUNE_TAIL2=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  x2=$t $v"
  done
  echo "$line"
done
