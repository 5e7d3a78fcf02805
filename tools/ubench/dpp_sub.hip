// dpp_sub.hip -- what v_sub_u32_dpp / v_subrev_u32_dpp with wave_shr:1 compute on gfx950, per lane, against what
// the ISA text says (V_SUB_U32: D = S0 - S1, V_SUBREV_U32: D = S1 - S0, DPP applies to S0).  Found in round 3: hipcc
// folds `base - dpp(x)` into v_subrev_u32_dpp and the column-wave kernel's window sums came out wrong.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
  const unsigned lane = threadIdx.x;
  unsigned x = 1000u * lane + 7u, base = 5000000u + lane;
  unsigned r_subrev, r_sub, r_mov, r_add;
  asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r_subrev) : "v"(x), "v"(base));
  asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r_sub) : "v"(x), "v"(base));
  asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r_mov) : "v"(x));
  asm volatile("s_nop 4\n\tv_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r_add) : "v"(x), "v"(base));
  out[lane * 4 + 0] = r_subrev; out[lane * 4 + 1] = r_sub; out[lane * 4 + 2] = r_mov; out[lane * 4 + 3] = r_add;
}
int main()
{
  unsigned* d; unsigned h[256];
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad_subrev = 0, bad_sub = 0, bad_add = 0;
  for (unsigned l = 0; l < 64; ++l) {
    const unsigned xl = l ? 1000u * (l - 1) + 7u : 0u, base = 5000000u + l;      // dpp(x): lane l-1's x, lane 0: 0
    if (h[l * 4 + 2] != xl) printf("mov lane %u: %u expected %u\n", l, h[l * 4 + 2], xl);
    bad_subrev += h[l * 4 + 0] != base - xl; bad_sub += h[l * 4 + 1] != xl - base; bad_add += h[l * 4 + 3] != xl + base;
  }
  for (unsigned l : {0u, 1u, 2u, 15u, 16u, 17u, 63u})
    printf("lane %2u: subrev_dpp %10u (S1-dpp(S0) = %10u, dpp(S0)-S1 = %10u)   sub_dpp %10u   mov_dpp %u   add_dpp %u\n", l, h[l * 4], 5000000u + l - (l ? 1000u * (l - 1) + 7u : 0u),
           (l ? 1000u * (l - 1) + 7u : 0u) - (5000000u + l), h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("lanes where v_subrev_u32_dpp != S1 - dpp(S0): %d; v_sub_u32_dpp != dpp(S0) - S1: %d; v_add_u32_dpp wrong: %d\n", bad_subrev, bad_sub, bad_add);
  return 0;
}
