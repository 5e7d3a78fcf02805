#!/bin/bash
# PMC pass for one bench_cfg.py configuration: VALU instruction count and busy cycles.
# usage (on the GPU box): tools/pmc_cfg.sh <tag> W H BLOCK MAXDISP NFRAMES
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS} --output-format csv -d $OUT -o pmc -- python3 $ROOT/tools/bench_cfg.py "$@" > $OUT/log.txt 2>&1
echo "exit=$?"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_colwave" in r["Kernel_Name"]:
        acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(dict)
for (d, c), v in acc.items(): per[d][c] = sum(v)
big = sorted(per.values(), key=lambda x: -x.get("GRBM_GUI_ACTIVE", 0))[:3]
for x in big:
    cyc = x["GRBM_GUI_ACTIVE"] / 8
    print({k: f"{v:.4g}" for k, v in x.items()}, "VALU busy %.1f%%" % (100 * x["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)), "ms at 2.4GHz %.3f" % (cyc / 2.4e6))
PY
