#!/bin/bash
# A/B of two builds of the product library: lib/ (new) against lib_prev/ (a copy of the previous build).  us per pair
cd "$(dirname "$0")/../.."
for c in "1920 1080 9 64 1" "1920 1080 16 64 1" "640 480 16 64 1" "640 480 7 32 1" "1280 720 9 64 1" "3840 2160 15 128 1" "1920 1080 9 64 4" "1920 1080 9 64 16" "1920 1080 9 64 128" "1920 1080 9 64 1024" \
         "1920 1080 16 64 64" "1920 1080 16 128 64" "1920 1080 16 96 64" "1920 1080 21 64 32" "1920 1080 16 80 64" "1920 1080 16 112 64" "1920 1080 9 112 64" "1920 1080 13 112 64" "1920 1080 21 80 32" "3840 2160 15 128 16" "640 480 7 32 128" "1920 1080 31 64 16"; do
  line="$c :"
  for t in prev new prev new; do
    lib=$PWD/steroscopic-hardware_amd/lib/libsadhip.so; [ $t = prev ] && lib=$PWD/steroscopic-hardware_amd/lib_prev/libsadhip.so
    v=$(SADHIP_LIB=$lib timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  $t $v"
  done
  echo "$line"
done
