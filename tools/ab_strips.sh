for c in "1920 1080 9 64 128" "1920 1080 9 64 16" "1920 1080 9 64 4" "1920 1080 9 64 1" "3840 2160 15 128 16" "3840 2160 15 128 4" "3840 2160 15 128 1" "1920 1080 16 64 64" "1920 1080 16 64 8" "1920 1080 16 64 1" "640 480 7 32 128" "640 480 7 32 8" "640 480 7 32 1" "1920 1080 15 64 64" "1920 1080 15 64 1" "1920 1080 11 64 1" "1920 1080 31 64 16" "1920 1080 3 16 64"; do
  a=$(SADHIP_TUNE_LEGACY_STRIPS=1 timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*/\1/')
  b=$(SADHIP_TUNE_VERBOSE=0 SADHIP_TUNE_ND=0 timeout -k 10 100 python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
  echo "$c : legacy $a  model $b"
done
