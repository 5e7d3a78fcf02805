#!/bin/bash
# usage: sweep_plan.sh W H BLOCK MAXDISP NFRAMES "nd:nw nd:nw ..."   (forces plans through SADHIP_TUNE_ND / _NW)
W=$1; H=$2; B=$3; D=$4; N=$5
for c in $6; do
  nd=${c%%:*}; nw=${c##*:}
  echo -n "ND=$nd NW=$nw  "
  SADHIP_TUNE_ND=$nd SADHIP_TUNE_NW=$nw python tools/bench_cfg.py $W $H $B $D $N
done
