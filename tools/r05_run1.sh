#!/bin/bash
# round 5, first GPU call: the suite once, then the host-path A/Bs and the scratch-pool microbenchmark
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest1.txt 2>&1; rc=$?
tail -5 $O/r05_pytest1.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/ab_hostcopy.py --quick > $O/r05_ab_hostcopy_quick.txt 2>&1 && tail -3 $O/r05_ab_hostcopy_quick.txt &&
timeout -k 10 200 python tools/ab_hostcopy.py --spin > $O/r05_ab_hostcopy_spin.txt 2>&1 && tail -7 $O/r05_ab_hostcopy_spin.txt &&
timeout -k 10 120 python tools/multipass_call.py > $O/r05_multipass_call.txt 2>&1 && cat $O/r05_multipass_call.txt &&
timeout -k 10 300 python bench.py > $O/r05_bench1.json 2> $O/r05_bench1.err; echo "bench rc=$?"; tail -c 600 $O/r05_bench1.json
