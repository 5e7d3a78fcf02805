#!/bin/bash
# tools/ab_h8.sh -- ON THE GPU BOX, round 4: the BlockSize-16/17 window sums (colwave_kernel.h stage_b, H == P == 8) as the
# committed three statements against ONE asm statement per candidate (28 s_nop per wave-step less).  Expects the two
# libraries as lib_r03/libsadhip_h8old.so and lib_r03/libsadhip_h8merged.so (built by hand from the two forms of the
# header; the merged form is quoted in DESIGN.md section 6b).  Result: profiles/r04_ab_h8_merged_asm.txt -- the merged form
# is 1-2 % SLOWER.
for rep in 1 2; do
for lib in h8old h8merged; do
  for d in 16 64 128; do
    SADHIP_LIB=lib_r03/libsadhip_$lib.so python tools/bench_cfg.py 1920 1080 16 $d 64 | tail -1
  done
done
done
