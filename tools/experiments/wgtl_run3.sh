#!/bin/bash
cd "$(dirname "$0")/../.."
for c in "1920 1080 9 64 1" "1920 1080 16 64 1" "640 480 16 64 1" "640 480 7 32 1" "3840 2160 15 128 1" "1920 1080 9 64 4"; do
  timeout -k 10 120 python tools/wg_timeline.py $c 2>&1 | grep -v amdgpu.ids
  SADHIP_TUNE_PRIO=0 timeout -k 10 120 python tools/wg_timeline.py $c 2>&1 | grep -v amdgpu.ids | sed 's/^/  [oldest-first only] /'
done
