for c in "1920 1080 16 192 64" "1920 1080 16 192 32" "1920 1080 16 256 64" "1920 1080 31 192 64" "1920 1080 9 192 64" "1920 1080 16 128 64" "1920 1080 9 64 128" "3840 2160 15 128 16" "1920 1080 16 64 64"; do
  a=$(SADHIP_LIB=steroscopic-hardware_amd/lib_r03/libsadhip.so python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
  b=$(python tools/bench_cfg.py $c 2>/dev/null | grep us/frame | sed 's/.*kind=1: \([0-9.]*\) us.*parity=\(.*\)/\1 \2/')
  echo "$c : r03 $a   r04 $b"
done
