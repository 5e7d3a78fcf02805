#!/bin/bash
# round 5, last GPU calls: the suite once on the final library, the profile set, the two-rank rehearsal of bench.py's N > 1 form, the grid
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest_final.txt 2>&1; rc=$?
tail -4 $O/r05_pytest_final.txt
[ $rc -eq 0 ] || exit $rc
bash tools/make_profiles.sh r05 2>&1 | grep -v "^|" | tail -25
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-one-gpu > $O/r05_rehearse2.json 2> $O/r05_rehearse2.err; echo "rehearse rc=$?"; tail -c 400 $O/r05_rehearse2.json
timeout -k 10 600 python tools/survey_grid_full.py $O/r05_grid_full.txt 64 > $O/r05_grid_full.log 2>&1; tail -3 $O/r05_grid_full.txt
