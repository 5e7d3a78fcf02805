for c in "640 480 7 32 1" "1920 1080 11 64 1" "3840 2160 15 128 1" "3840 2160 15 128 4" "1920 1080 9 64 1"; do
  for i in 1 2 3; do
  a=$(SADHIP_TUNE_LEGACY_STRIPS=1 SADHIP_TUNE_VERBOSE=1 timeout -k 10 100 python tools/bench_cfg.py $c 2>&1 | grep -E "us/frame|strips of" | tail -2 | sed 's/.*kind=1: \([0-9.]*\) us.*/\1/; s/sadhip: //' | tr '\n' ' ')
  b=$(SADHIP_TUNE_VERBOSE=1 timeout -k 10 100 python tools/bench_cfg.py $c 2>&1 | grep -E "us/frame|strips of" | tail -2 | sed 's/.*kind=1: \([0-9.]*\) us.*/\1/; s/sadhip: //' | tr '\n' ' ')
  echo "$c : LEGACY $a | MODEL $b"
  done
done
