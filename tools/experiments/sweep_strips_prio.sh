#!/bin/bash
# Round 5: strip counts for launches of kernels that fit one or two workgroups per CU (8- and 16-wave workgroups): the model's
# choice against 1.5x / 2x / 3x / 4x as many (equal) strips.  us per pair; every run is its own process (the tuning knobs are
# read once).  usage (GPU box): tools/sweep_strips_bigwg.sh > gpurun_out/r05_sweep_strips_bigwg.txt
cd "$(dirname "$0")/../.."
geo() { python - "$@" <<'PY'
import sys
sys.path.insert(0, ".")
import torch
import steroscopic_hardware_amd as sha
torch.zeros(1, device="cuda")
w, h, b, d, n = map(int, sys.argv[1:6])
g = sha.capi.plan_describe_batch(w, h, b, d, n)
print(" ".join(f"{p['strips']}:{p['wg_per_cu']}:{p['nw']}" for p in g))
PY
}
for c in "1920 1080 9 64 1" "1920 1080 9 64 4" "1920 1080 9 64 16" "1920 1080 9 64 48" "1920 1080 9 64 128" "1920 1080 16 64 1" "1920 1080 16 64 8" "1920 1080 16 64 64" "3840 2160 15 128 1" "3840 2160 15 128 16" "640 480 7 32 16" "1280 720 9 64 8" "1920 1080 15 64 64" "1920 1080 31 64 16"; do
  g=$(geo $c 2>/dev/null | tail -1)
  s0=${g%%:*}
  line="$c : plan(strips:wg_per_cu:waves) $g |"
  for m in 50 75 100 150 200; do
    s=$(( (s0 * m + 50) / 100 ))
    v=$(SADHIP_TUNE_STRIPS=$s timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
    line="$line  strips=$s $v"
  done
  echo "$line"
done
