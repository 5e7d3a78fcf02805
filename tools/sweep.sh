#!/bin/bash
# tuning sweep on the GPU box: prints ms/frame for combinations of chunk size and workgroup target
cd ${GRAFT_REPO_ROOT:-.}
for nd in 8 12 16; do for wgs in 1024 2048 4096 8192; do
  r=$(SADHIP_TUNE_ND=$nd SADHIP_TUNE_WGS=$wgs timeout -k 5 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --frames-per-gpu 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/frame  single %.4f  %.0f Gcand/s' % (d['ms_per_frame'], d['ms_per_frame_single_pair_latency'], d['gcand_per_s']))")
  echo "ND=$nd WGS=$wgs : $r"
done; done
