#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for dbg in 0 1 2 4 8 3 7 15 6 14; do
  r=$(SADHIP_TUNE_DBG=$dbg timeout -k 5 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --frames-per-gpu 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/frame  single %.4f' % (d['ms_per_frame'], d['ms_per_frame_single_pair_latency']))")
  echo "DBG=$dbg : $r"
done
