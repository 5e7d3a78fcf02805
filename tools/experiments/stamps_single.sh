#!/bin/bash
cd "$(dirname "$0")/../.."
export SADHIP_LIB=$PWD/steroscopic-hardware_amd/lib/libsadhip_stamps.so
for c in "1920 1080 9 64 1" "1920 1080 9 64 128" "1920 1080 16 64 1" "640 480 16 64 1"; do
  echo "== $c"; SADHIP_TUNE_STAMPS=1 timeout -k 10 100 python tools/bench_cfg.py $c 2>&1 | grep "sadhip stamps\|us/frame" | sort | uniq -c | sort -rn | head -6
done
