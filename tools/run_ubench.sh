#!/bin/bash
# tools/run_ubench.sh [outdir] -- ON THE GPU BOX: runs every microbenchmark the roofline argument of DESIGN.md section 4.1
# stands on and keeps its complete printed output, stamped with the sha256 of the source it was built from:
#   <outdir>/ubench_<name>.txt     (copied into profiles/ afterwards)
#   <outdir>/ubench_dpp_sub_disasm.txt   llvm-objdump's view (mnemonic, operands, encoding) of the DPP subtract
#                                        instructions dpp_sub executes
# Builds what is missing with hipcc (seconds per file).  Each program is bounded by `timeout -k 10 120`.
set -u
here="$(cd "$(dirname "$0")" && pwd)"; ub="$here/ubench"
out="${1:-gpurun_out/ubench}"; mkdir -p "$out"
HIPCC=/opt/rocm/bin/hipcc
rc=0
for name in ${UBENCH_ONLY:-valu_rates valu_rates2 valu_mix valu_runs block_bench pcie_bw dpp_sub hbm_calib simd_map bank_test malloc_async buffer_range h2d_paths pin_reuse dpp_hazard}; do
  src="$ub/$name.hip"; bin="$ub/$name"
  if [ ! -x "$bin" ] || [ "$src" -nt "$bin" ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-command-line-argument -Wno-unused-result -o "$bin" "$src" 2> "$out/build_$name.log" || { echo "build of $name failed"; rc=1; continue; }
  fi
  f="$out/ubench_$name.txt"
  {
    echo "# tools/ubench/$name.hip  sha256 $(sha256sum "$src" | cut -c1-16)  $(date -u +%Y-%m-%dT%H:%M:%SZ)"
    echo "# $($HIPCC --version | grep -m1 -i 'hip version')"
  } > "$f"
  timeout -k 10 120 "$bin" >> "$f" 2>&1 || { echo "# exit status $?" >> "$f"; echo "$name: non-zero exit"; rc=1; }
  echo "$name: $(wc -l < "$f") lines"
done
# the encoding of the instructions dpp_sub runs
if $HIPCC --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output -Wno-unused-command-line-argument -c -o "$out/dpp_sub.co" "$ub/dpp_sub.hip" 2>/dev/null; then
  {
    echo "# llvm-objdump -d of tools/ubench/dpp_sub.hip (sha256 $(sha256sum "$ub/dpp_sub.hip" | cut -c1-16))"
    echo "# VOP2 word: [30:25] opcode (52 v_add_u32, 53 v_sub_u32, 54 v_subrev_u32), [24:17] vdst, [16:9] vsrc1, [8:0] src0 = 0xFA (DPP)"
    echo "# DPP word:  [7:0] the VGPR src0 reads, [16:8] dpp_ctrl (0x138 wave_shr:1, 0x130 wave_shl:1, 0x111 row_shr:1), [19] bound_ctrl"
    /opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 "$out/dpp_sub.co" | grep -E "_dpp|^[0-9a-f]+ <"
  } > "$out/ubench_dpp_sub_disasm.txt"
fi
rm -f "$out/dpp_sub.co"
exit $rc
