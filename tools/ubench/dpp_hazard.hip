// dpp_hazard.hip -- which operand of a DPP instruction needs the two wait states after the VALU instruction that wrote it?
// The ISA text says "VALU writes VGPR -> DPP reads that VGPR: 2 wait states"; LLVM's hazard recognizer applies that to EVERY
// VGPR the DPP instruction reads (the lane-shifted src0, the plain src1, the old value).  The window sums of the
// BlockSize-16/17 kernel (colwave_kernel.h, H == P == 8) use a value written by the instruction before as the PLAIN
// operand (src1) of v_sub_u32_dpp / v_add_u32_dpp seven + one times per candidate; padding those with s_nop costs an issue
// slot each.  This program runs the sequences with 0, 1 and 2 wait states in front of the DPP instruction, with the fresh
// register as src0 (lane-shifted) and as src1 (plain), and counts wrong lanes against values computed without DPP
// (__shfl_up / __shfl_down), over many workgroups, iterations and data patterns, with 1 and with 8 waves per SIMD resident
// (a lone wave issues back to back: the worst case for a missing wait state).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

#define CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// the fresh register as the lane-shifted operand (src0): y = shr(x) - c, x = p + q written NOPS wait states before
#define SRC0(NAME, NOPS)                                                                                      \
  __device__ __forceinline__ unsigned NAME(unsigned p, unsigned q, unsigned c)                                \
  {                                                                                                           \
    unsigned x = 0xdead0000u, y;                                                                              \
    asm volatile("v_mov_b32 %1, 0xdead0000\n\ts_nop 4\n\t"                                                    \
                 "v_add_u32 %1, %3, %4\n\t" NOPS                                                              \
                 "v_sub_u32_dpp %0, %1, %2 wave_shr:1" CTRL "s_nop 4"                                         \
                 : "=&v"(y), "+&v"(x) : "v"(c), "v"(p), "v"(q));                                              \
    return y;                                                                                                 \
  }
// ... as the plain operand (src1): y = shr(c) - x
#define SRC1(NAME, NOPS)                                                                                      \
  __device__ __forceinline__ unsigned NAME(unsigned p, unsigned q, unsigned c)                                \
  {                                                                                                           \
    unsigned x = 0xdead0000u, y;                                                                              \
    asm volatile("v_mov_b32 %1, 0xdead0000\n\ts_nop 4\n\t"                                                    \
                 "v_add_u32 %1, %3, %4\n\t" NOPS                                                              \
                 "v_sub_u32_dpp %0, %2, %1 wave_shr:1" CTRL "s_nop 4"                                         \
                 : "=&v"(y), "+&v"(x) : "v"(c), "v"(p), "v"(q));                                              \
    return y;                                                                                                 \
  }
SRC0(src0_w0, "")
SRC0(src0_w1, "s_nop 0\n\t")
SRC0(src0_w2, "s_nop 1\n\t")
SRC1(src1_w0, "")
SRC1(src1_w1, "s_nop 0\n\t")
SRC1(src1_w2, "s_nop 1\n\t")

// the chain the kernel would run per candidate: ps8 fresh -> (2 wait states) -> base = ps8 + shr(ps8) -> n1 = shr(ps1) - base
// (base written by the DPP instruction right before, plain operand) -> s0 = shl(ps1) + base -> w1 = shl(ps2) - n1
struct Chain { unsigned base, n1, s0, w1; };
__device__ __forceinline__ Chain chain(unsigned ps7, unsigned c7, unsigned ps1, unsigned ps2)
{
  Chain r; unsigned ps8;
  asm volatile("v_add_u32 %4, %5, %6\n\t"
               "s_nop 1\n\t"
               "v_add_u32_dpp %0, %4, %4 wave_shr:1" CTRL
               "v_sub_u32_dpp %1, %7, %0 wave_shr:1" CTRL
               "v_add_u32_dpp %2, %7, %0 wave_shl:1" CTRL
               "v_sub_u32_dpp %3, %8, %1 wave_shl:1" CTRL "s_nop 4"
               : "=&v"(r.base), "=&v"(r.n1), "=&v"(r.s0), "=&v"(r.w1), "=&v"(ps8)
               : "v"(ps7), "v"(c7), "v"(ps1), "v"(ps2));
  return r;
}

__global__ void k_hazard(unsigned long long* bad, int iters)
{
  const unsigned lane = threadIdx.x & 63u;
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const unsigned p = mix(id * 4u + (unsigned)it * 0x9e3779b9u), q = mix(p + 1u), c = mix(p + 2u), e = mix(p + 3u);
    const unsigned x = p + q;
    const unsigned xs = __shfl_up(x, 1, 64), cs = __shfl_up(c, 1, 64);
    const unsigned want0 = (lane ? xs : 0u) - c;                    // shr(x) - c
    const unsigned want1 = (lane ? cs : 0u) - x;                    // shr(c) - x
    b[0] += src0_w0(p, q, c) != want0;
    b[1] += src0_w1(p, q, c) != want0;
    b[2] += src0_w2(p, q, c) != want0;
    b[3] += src1_w0(p, q, c) != want1;
    b[4] += src1_w1(p, q, c) != want1;
    b[5] += src1_w2(p, q, c) != want1;
    // chain: ps7 = p, c7 = q, ps1 = c, ps2 = e
    const Chain r = chain(p, q, c, e);
    const unsigned ps8 = p + q;
    const unsigned l8 = __shfl_up(ps8, 1, 64), l1 = __shfl_up(c, 1, 64);
    const unsigned r1 = __shfl_down(c, 1, 64), r2 = __shfl_down(e, 1, 64);
    const unsigned base = ps8 + (lane ? l8 : 0u);
    const unsigned n1 = (lane ? l1 : 0u) - base;
    const unsigned s0 = (lane < 63 ? r1 : 0u) + base;
    const unsigned w1 = (lane < 63 ? r2 : 0u) - n1;
    b[6] += (r.base != base) + (r.n1 != n1) + (r.s0 != s0) + (r.w1 != w1);
  }
  for (int k = 0; k < 7; ++k)
    if (b[k]) atomicAdd(&bad[k], b[k]);
}

int main()
{
  unsigned long long* d; unsigned long long h[7];
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return 1;
  const char* names[7] = {"fresh register = lane-shifted operand (src0), 0 wait states", "   ... 1 wait state (s_nop 0)", "   ... 2 wait states (s_nop 1)",
                          "fresh register = plain operand (src1),        0 wait states", "   ... 1 wait state (s_nop 0)", "   ... 2 wait states (s_nop 1)",
                          "chain base -> n1 / s0 -> w1 (plain operands written by the DPP instruction before), 4 values"};
  struct { int blocks, threads, iters; const char* what; } runs[] = {
      {1024, 64, 2000, "1 wave per workgroup, 1024 workgroups (1 wave per SIMD: back-to-back issue)"},
      {8192, 256, 500, "4 waves per workgroup, 8192 workgroups (8 waves per SIMD resident)"},
  };
  int rc = 0;
  for (auto& r : runs) {
    if (hipMemset(d, 0, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_hazard, dim3(r.blocks), dim3(r.threads), 0, 0, d, r.iters);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const double n = (double)r.blocks * r.threads * r.iters;
    printf("%s: %.3g lane-evaluations per sequence\n", r.what, n);
    for (int k = 0; k < 7; ++k) printf("  %-95s wrong lanes: %llu\n", names[k], h[k]);
    if (h[2] || h[5]) rc = 1;                                          // two wait states must always be right
  }
  return rc;
}
