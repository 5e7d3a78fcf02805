#!/bin/bash
cd "$(dirname "$0")/../.."
for c in "3840 2160 15 128 16" "1920 1080 16 64 64" "1920 1080 9 64 128" "1920 1080 9 64 16"; do
  timeout -k 10 120 python tools/wg_timeline.py $c 2>&1 | grep -v amdgpu.ids
  SADHIP_TUNE_PRIO=0 timeout -k 10 120 python tools/wg_timeline.py $c 2>&1 | grep -v amdgpu.ids | sed 's/^/  [oldest-first only] /'
done
