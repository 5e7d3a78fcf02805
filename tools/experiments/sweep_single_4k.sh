#!/bin/bash
cd "$(dirname "$0")/../.."
run() { timeout -k 10 100 python tools/bench_cfg.py "$@" 2>/dev/null | grep us/frame; }
echo "== 4K 15/128 single pair: model, then forced strip counts"; SADHIP_TUNE_VERBOSE=1 python tools/bench_cfg.py 3840 2160 15 128 1 2>&1 | grep "sadhip:" | sort | uniq -c | head -3
run 3840 2160 15 128 1
for s in 16 24 32 40 48 64 96; do echo -n "strips=$s "; SADHIP_TUNE_STRIPS=$s run 3840 2160 15 128 1; done
echo "== 4K 15/128 two pairs"; run 3840 2160 15 128 2
for s in 16 24 32 48; do echo -n "strips=$s "; SADHIP_TUNE_STRIPS=$s run 3840 2160 15 128 2; done
echo "== 1080p 15/128 single"; run 1920 1080 15 128 1
for s in 16 24 32 48 64; do echo -n "strips=$s "; SADHIP_TUNE_STRIPS=$s run 1920 1080 15 128 1; done
echo "== 1080p 16/64 two pairs"; run 1920 1080 16 64 2
for s in 30 45 60 90; do echo -n "strips=$s "; SADHIP_TUNE_STRIPS=$s run 1920 1080 16 64 2; done
