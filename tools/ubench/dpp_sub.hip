// dpp_sub.hip -- what the DPP forms of the 32-bit subtract instructions compute per lane on gfx950, classified against the
// four readings one could have of "D = v_sub*_u32_dpp S0, S1 <lane shift>":
//     A  shift(S0) - S1        B  S1 - shift(S0)        C  S0 - shift(S1)        D  shift(S1) - S0
// The ISA text says V_SUB_U32: D = S0 - S1, V_SUBREV_U32: D = S1 - S0, DPP selects the lane of S0 -> sub_dpp = A, subrev_dpp = B.
// Round 3: the column-wave kernel's window sums for BlockSize 16 were wrong on every pixel after hipcc folded
// `base - dpp(x)` (v_mov_b32_dpp + v_sub_u32) into one v_subrev_u32_dpp; this program is the isolated check.  It prints,
// per instruction form and lane-shift control, how many of the 64 lanes agree with each reading, plus a few raw lanes.
// The operand ORDER the assembler gives the instruction (which register lands in the src0 field, the one DPP applies to)
// is a separate question: `make -C tools/ubench dpp_sub.dis` / profiles/ubench_dpp_sub_disasm.txt holds llvm-objdump's view
// of the very instructions executed here.
#include <hip/hip_runtime.h>
#include <cstdio>

#define DPPK(name, ins, ctrl)                                                                                              \
  __global__ void name(unsigned* out)                                                                                     \
  {                                                                                                                       \
    const unsigned lane = threadIdx.x;                                                                                    \
    unsigned a = 1000u * lane + 7u, b = 5000000u + 3u * lane, r;                                                          \
    asm volatile("s_nop 4\n\t" ins " %0, %1, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=&v"(r) : "v"(a), "v"(b)); \
    out[lane] = r;                                                                                                        \
  }
// first asm operand after the destination = a ("S0" as written), second = b ("S1" as written)
DPPK(k_sub_wshr, "v_sub_u32_dpp", "wave_shr:1")
DPPK(k_subrev_wshr, "v_subrev_u32_dpp", "wave_shr:1")
DPPK(k_sub_rshr, "v_sub_u32_dpp", "row_shr:1")
DPPK(k_subrev_rshr, "v_subrev_u32_dpp", "row_shr:1")
DPPK(k_add_wshr, "v_add_u32_dpp", "wave_shr:1")
DPPK(k_sub_wshl, "v_sub_u32_dpp", "wave_shl:1")
DPPK(k_subrev_wshl, "v_subrev_u32_dpp", "wave_shl:1")

// what the compiler makes of `b - dpp(a)` and `dpp(a) - b` written with the builtin (the fold that went wrong)
__global__ void k_builtin(unsigned* out)
{
  const unsigned lane = threadIdx.x;
  unsigned a = 1000u * lane + 7u, b = 5000000u + 3u * lane;
  const unsigned sa = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
  out[lane] = b - sa;
  out[64 + lane] = sa - b;
}

static unsigned A_(unsigned l) { return 1000u * l + 7u; }
static unsigned B_(unsigned l) { return 5000000u + 3u * l; }

// src lane of destination lane l under the control, or -1 (bound_ctrl:1 -> the operand reads 0)
static int src_lane(const char* ctrl, int l)
{
  if (ctrl[0] == 'w' && ctrl[7] == 'r') return l - 1;                              // wave_shr:1
  if (ctrl[0] == 'w') return l + 1 < 64 ? l + 1 : -1;                              // wave_shl:1
  return (l % 16) ? l - 1 : -1;                                                    // row_shr:1
}

static void classify(const char* name, const char* ctrl, const unsigned* h, bool add)
{
  int nA = 0, nB = 0, nC = 0, nD = 0;
  for (int l = 0; l < 64; ++l) {
    const int s = src_lane(ctrl, l);
    const unsigned sa = s < 0 ? 0u : A_(s), sb = s < 0 ? 0u : B_(s), a = A_(l), b = B_(l);
    if (add) { nA += h[l] == sa + b; nC += h[l] == a + sb; continue; }
    nA += h[l] == sa - b; nB += h[l] == b - sa; nC += h[l] == a - sb; nD += h[l] == sb - a;
  }
  if (add) printf("%-28s %-10s lanes equal to  shift(S0)+S1: %2d   S0+shift(S1): %2d\n", name, ctrl, nA, nC);
  else printf("%-28s %-10s lanes equal to  A shift(S0)-S1: %2d   B S1-shift(S0): %2d   C S0-shift(S1): %2d   D shift(S1)-S0: %2d\n", name, ctrl, nA, nB, nC, nD);
  printf("    raw lanes 0,1,2,16,17,63: %u %u %u %u %u %u\n", h[0], h[1], h[2], h[16], h[17], h[63]);
}

#define RUN(kern, name, ctrl, add)                                                             \
  do {                                                                                         \
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d);                                      \
    if (hipMemcpy(h, d, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return 1; \
    classify(name, ctrl, h, add);                                                              \
  } while (0)

int main()
{
  unsigned* d; unsigned h[128];
  if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return 1;
  printf("operands as written in the asm: S0 = a = 1000*lane+7, S1 = b = 5000000+3*lane; bound_ctrl:1 (a lane without source reads 0)\n");
  RUN(k_sub_wshr, "v_sub_u32_dpp D, a, b", "wave_shr:1", false);
  RUN(k_subrev_wshr, "v_subrev_u32_dpp D, a, b", "wave_shr:1", false);
  RUN(k_sub_rshr, "v_sub_u32_dpp D, a, b", "row_shr:1", false);
  RUN(k_subrev_rshr, "v_subrev_u32_dpp D, a, b", "row_shr:1", false);
  RUN(k_sub_wshl, "v_sub_u32_dpp D, a, b", "wave_shl:1", false);
  RUN(k_subrev_wshl, "v_subrev_u32_dpp D, a, b", "wave_shl:1", false);
  RUN(k_add_wshr, "v_add_u32_dpp D, a, b", "wave_shr:1", true);
  hipLaunchKernelGGL(k_builtin, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int ok1 = 0, ok2 = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned sa = l ? A_(l - 1) : 0u;
    ok1 += h[l] == B_(l) - sa; ok2 += h[64 + l] == sa - B_(l);
  }
  printf("compiler, C source `b - update_dpp(a, wave_shr:1)`: %2d of 64 lanes right;  `update_dpp(a) - b`: %2d of 64 right\n", ok1, ok2);
  return 0;
}
