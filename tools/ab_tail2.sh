#!/bin/bash
# Round 5 A/B of the strip count of kernels that fit two workgroups per CU, launches of more than one round: SADHIP_TUNE_TAIL2=0 (the model's
# count, rounds 2-4) against the default (twice the model's count; choose_strips in colwave_plan.hip).  us per pair, alternating, every run its
# own process.  profiles/r05_ab_strips_x2.txt is this script's output; profiles/r05_ab_tail2_first.txt the same cases with a tail term in the
# model instead (ideal rounds + half a workgroup: it picked ONE round of very tall workgroups for several cases and lost 3-7 % there).
# usage (GPU box): tools/ab_tail2.sh > gpurun_out/r05_ab_tail2.txt
cd "$(dirname "$0")/.."
for c in "3840 2160 15 128 4" "3840 2160 15 128 8" "3840 2160 15 128 16" "3840 2160 15 128 32" "3840 2160 15 128 1" "1920 1080 15 128 16" "1920 1080 15 128 64" "1920 1080 13 128 32" "1920 1080 11 96 32" \
         "1920 1080 16 64 4" "1920 1080 16 64 16" "1920 1080 16 64 64" "1920 1080 16 64 256" "1920 1080 17 32 64" "1920 1080 9 128 64" "1920 1080 9 128 8" "1920 1080 16 128 64" "1920 1080 9 64 128" "640 480 16 64 64" "1280 720 15 128 16"; do
  line="$c :"
  for t in 0 1 0 1; do
    v=$(SADHIP_TUNE_TAIL2=$t timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  tail2=$t $v"
  done
  echo "$line"
done
