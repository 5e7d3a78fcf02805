// valu_rates.hip -- measures the issue rate of the integer VALU instructions the SAD
// disparity kernels are built from, on the GPU it runs on (gfx950).  Diagnostic tool only;
// its numbers are recorded in DESIGN.md ("VALU ceiling").
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
//
// Each kernel runs ITER x 16 independent instances of one instruction per lane (8 chains x 2),
// with 4 waves/SIMD resident on every CU, and reports lane-ops/s and the implied cycles per
// wave-instruction per SIMD at the clock reported by hipDeviceProp (nominal; see note printed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;

// 32-bit accumulator ops: "OP acc, a, b, acc"-style strings use %0 (acc, in/out) %1 %2 (inputs)
#define KERNEL32(NAME, ASM)                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned va = a ^ threadIdx.x, vb = b + threadIdx.x * 3u;                      \
  unsigned c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3,              \
           c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;                   \
  for (int i = 0; i < ITER; ++i) {                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                              \
      asm volatile(ASM : "+v"(c0) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c1) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c2) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c3) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c4) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c5) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c6) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c7) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
    }                                                                            \
  }                                                                              \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; \
}

// 64-bit accumulator ops (qsad family): %0 = v[lo:hi] acc, %1 = 64-bit src, %2 = 32-bit src
#define KERNEL64(NAME, ASM)                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned long long va = ((unsigned long long)(a ^ threadIdx.x) << 32) | (b * 7u + threadIdx.x); \
  unsigned vb = b + threadIdx.x * 3u;                                            \
  unsigned long long c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3,    \
           c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;                   \
  for (int i = 0; i < ITER; ++i) {                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                              \
      asm volatile(ASM : "+v"(c0) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c1) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c2) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c3) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c4) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c5) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c6) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
      asm volatile(ASM : "+v"(c7) : "v"(va), "v"(vb) : "vcc", "s20", "s21");                           \
    }                                                                            \
  }                                                                              \
  unsigned long long r = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32);\
}

KERNEL32(k_add_u32,      "v_add_u32 %0, %1, %0")
KERNEL32(k_sub_u32,      "v_sub_u32 %0, %0, %1")
KERNEL32(k_sad_u8,       "v_sad_u8 %0, %1, %2, %0")
KERNEL32(k_sad_hi_u8,    "v_sad_hi_u8 %0, %1, %2, %0")
KERNEL32(k_sad_u16,      "v_sad_u16 %0, %1, %2, %0")
KERNEL32(k_sad_u32,      "v_sad_u32 %0, %1, %2, %0")
KERNEL32(k_msad_u8,      "v_msad_u8 %0, %1, %2, %0")
KERNEL32(k_pk_add_u16,   "v_pk_add_u16 %0, %1, %0")
KERNEL32(k_pk_sub_u16,   "v_pk_sub_u16 %0, %0, %1")
KERNEL32(k_pk_min_u16,   "v_pk_min_u16 %0, %1, %0")
KERNEL32(k_pk_max_u16,   "v_pk_max_u16 %0, %1, %0")
KERNEL32(k_pk_sub_i16,   "v_pk_sub_i16 %0, %0, %1")
KERNEL32(k_pk_mad_u16,   "v_pk_mad_u16 %0, %1, %2, %0")
KERNEL32(k_min_u32,      "v_min_u32 %0, %1, %0")
KERNEL32(k_min3_u32,     "v_min3_u32 %0, %1, %2, %0")
KERNEL32(k_perm_b32,     "v_perm_b32 %0, %1, %0, %2")
KERNEL32(k_alignbyte,    "v_alignbyte_b32 %0, %1, %0, %2")
KERNEL32(k_bfe_u32,      "v_bfe_u32 %0, %0, 8, 8")
KERNEL32(k_lshl_or,      "v_lshl_or_b32 %0, %0, 16, %1")
KERNEL32(k_and_or,       "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_add3_u32,     "v_add3_u32 %0, %1, %2, %0")
KERNEL32(k_lshl_add,     "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_xad_u32,      "v_xad_u32 %0, %0, %1, %2")
KERNEL32(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_add_dpp_rshr1,"v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_rshr4,"v_add_u32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_wshr1,"v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_wshl1,"v_add_u32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp_bc15, "v_add_u32_dpp %0, %1, %0 row_bcast:15 row_mask:0xf bank_mask:0xf")
KERNEL32(k_mov_dpp_rshr1,"v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_sub_sdwa,     "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(k_dot4_u32_u8,  "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL32(k_max3_u32,     "v_max3_u32 %0, %1, %2, %0")
KERNEL32(k_med3_u32,     "v_med3_u32 %0, %1, %2, %0")

KERNEL32(k_and_b32,      "v_and_b32 %0, %1, %0")
KERNEL32(k_or_b32,       "v_or_b32 %0, %1, %0")
KERNEL32(k_xor_b32,      "v_xor_b32 %0, %1, %0")
KERNEL32(k_lshlrev_b32,  "v_lshlrev_b32 %0, 1, %0")
KERNEL32(k_lshrrev_b32,  "v_lshrrev_b32 %0, 1, %0")
KERNEL32(k_max_u32,      "v_max_u32 %0, %1, %0")
KERNEL32(k_mov_b32,      "v_mov_b32 %0, %1")
KERNEL32(k_add_u16,      "v_add_u16 %0, %1, %0")
KERNEL32(k_min_u16,      "v_min_u16 %0, %1, %0")
KERNEL32(k_mul_u32_u24,  "v_mul_u32_u24 %0, %1, %0")
KERNEL32(k_add_f32,      "v_add_f32 %0, %1, %0")
KERNEL32(k_fma_f32,      "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_add_u32_e64,  "v_add_u32_e64 %0, %1, %0")
KERNEL32(k_add_co_u32,   "v_add_co_u32 %0, vcc, %1, %0")
KERNEL32(k_add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL32(k_add_lshl,     "v_add_lshl_u32 %0, %0, %1, 1")
KERNEL32(k_cmp_cnd,      "v_cmp_lt_u32 s[20:21], %1, %0\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cnd_e64,      "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cmp_only,     "v_cmp_lt_u32 s[20:21], %1, %0\n\tv_add_u32 %0, %1, %0")
KERNEL32(k_mix_add_sad,  "v_add_u32 %0, %1, %0\n\tv_sad_u8 %0, %1, %2, %0")
KERNEL32(k_mix_add_add_sad, "v_add_u32 %0, %1, %0\n\tv_sub_u32 %0, %0, %2\n\tv_sad_u8 %0, %1, %2, %0")
KERNEL32(k_pk_lshl_b16,  "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL32(k_pk_max_i16,   "v_pk_max_i16 %0, %1, %0")
KERNEL32(k_sub_u32_sgpr, "v_sub_u32 %0, %0, s20")
KERNEL32(k_sad_u8_sgpr,  "v_sad_u8 %0, s20, %2, %0")
KERNEL32(k_addc_u32,     "v_addc_co_u32 %0, vcc, %1, %0, vcc")
// same instructions with 24 independent chains per lane (latency vs issue rate)
#define KERNEL32W(NAME, ASM)                                                     \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned va = a ^ threadIdx.x, vb = b + threadIdx.x * 3u;                      \
  unsigned c[24];                                                                \
  _Pragma("unroll") for (int k = 0; k < 24; ++k) c[k] = threadIdx.x + k;         \
  for (int i = 0; i < ITER * 16 / 24; ++i) {                                     \
    _Pragma("unroll") for (int k = 0; k < 24; ++k)                               \
      asm volatile(ASM : "+v"(c[k]) : "v"(va), "v"(vb) : "vcc", "s20", "s21");  \
  }                                                                              \
  unsigned r = 0;                                                                \
  _Pragma("unroll") for (int k = 0; k < 24; ++k) r ^= c[k];                      \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                \
}
KERNEL32W(kw_add_u32,    "v_add_u32 %0, %1, %0")
KERNEL32W(kw_sad_u32,    "v_sad_u32 %0, %1, %2, %0")
KERNEL32W(kw_sad_u8,     "v_sad_u8 %0, %1, %2, %0")
KERNEL32W(kw_min3_u32,   "v_min3_u32 %0, %1, %2, %0")
KERNEL32W(kw_min_u32,    "v_min_u32 %0, %1, %0")
KERNEL32W(kw_add_dpp,    "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
KERNEL32W(kw_pk_add_u16, "v_pk_add_u16 %0, %1, %0")
KERNEL32W(kw_add3_u32,   "v_add3_u32 %0, %1, %2, %0")
KERNEL32W(kw_cndmask,    "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32W(kw_sad_const,  "v_sad_u32 %0, %1, %2, 0")
KERNEL32W(kw_sub_sgpr,   "v_sub_u32 %0, %0, s20")
KERNEL64(k_qsad_pk,      "v_qsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL64(k_mqsad_pk,     "v_mqsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL64(k_lshl_b64,     "v_lshlrev_b64 %0, 1, %0")

// 128-bit accumulator: v_mqsad_u32_u8 v[0:3], v[4:5], v6, v[0:3]
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_mqsad_u32(unsigned* out, unsigned a, unsigned b) {
  unsigned long long va = ((unsigned long long)(a ^ threadIdx.x) << 32) | (b * 7u + threadIdx.x);
  unsigned vb = b + threadIdx.x * 3u;
  u32x4 c0 = {threadIdx.x, 1, 2, 3}, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(c0) : "v"(va), "v"(vb) : "vcc", "s20", "s21");
      asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(c1) : "v"(va), "v"(vb) : "vcc", "s20", "s21");
      asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(c2) : "v"(va), "v"(vb) : "vcc", "s20", "s21");
      asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(c3) : "v"(va), "v"(vb) : "vcc", "s20", "s21");
    }
  }
  u32x4 r = c0 ^ c1 ^ c2 ^ c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.x ^ r.y ^ r.z ^ r.w;
}

// ---- semantics probes -------------------------------------------------------------------
__global__ void k_semantics(unsigned long long* o) {
  // qsad: accumulator lane wrap or saturate?  S0 bytes 0xFF.., S1 = 0 -> each SAD = 4*255 = 1020
  unsigned long long s0 = 0xFFFFFFFFFFFFFFFFull; unsigned s1 = 0;
  unsigned long long acc = 0xFFF0FFF0FFF0FFF0ull;           // 65520 per lane: +1020 overflows u16
  o[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc);
  // qsad byte/shift order: S0 = bytes 00 01 02 .. 07 (byte0 = LSB), S1 = bytes 10 20 30 40
  o[1] = __builtin_amdgcn_qsad_pk_u16_u8(0x0706050403020100ull, 0x40302010u, 0ull);
  // sad_u8 carries between halves? acc low half 0xFFFF + 1020
  o[2] = __builtin_amdgcn_sad_u8(0xFFFFFFFFu, 0u, 0x0001FFFFu);
  o[3] = __builtin_amdgcn_sad_hi_u8(0xFFFFFFFFu, 0u, 0xFFFF0005u);
  // msad: which operand's zero byte masks?  (a) S0 has a zero byte  (b) S1 has a zero byte
  o[4] = __builtin_amdgcn_msad_u8(0x05000505u, 0x01010101u, 0u);
  o[5] = __builtin_amdgcn_msad_u8(0x05050505u, 0x01000101u, 0u);
  // mqsad_pk same data as o[1]
  o[6] = __builtin_amdgcn_mqsad_pk_u16_u8(0x0706050403020100ull, 0x40302010u, 0ull);
  o[7] = __builtin_amdgcn_mqsad_pk_u16_u8(0x0706050403020100ull, 0x40300010u, 0ull);
  // DPP wave_shr:1 / wave_shl:1 availability + semantics: lane id shifted
  unsigned lane = threadIdx.x;
  unsigned w1 = __builtin_amdgcn_update_dpp(999u, lane, 0x138, 0xf, 0xf, false); // wave_shr:1
  unsigned w2 = __builtin_amdgcn_update_dpp(999u, lane, 0x130, 0xf, 0xf, false); // wave_shl:1
  unsigned r1 = __builtin_amdgcn_update_dpp(999u, lane, 0x111, 0xf, 0xf, false); // row_shr:1
  unsigned r1z = __builtin_amdgcn_update_dpp(999u, lane, 0x111, 0xf, 0xf, true); // row_shr:1 bound_ctrl
  o[8 + lane] = ((unsigned long long)w1 << 48) | ((unsigned long long)w2 << 32) | ((unsigned long long)r1 << 16) | r1z;
}

struct Entry { const char* name; void (*fn)(unsigned*, unsigned, unsigned); double ops_per_inst; };

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount; double clk = p.clockRate * 1e3;
  printf("device: %s  CUs=%d  clockRate=%.0f MHz  arch=%s\n", p.name, cus, clk / 1e6, p.gcnArchName);
  const int blocks = cus * 16, threads = 256;   // 4 blocks x 4 waves resident per CU (<=64 VGPR)
  unsigned* d; CK(hipMalloc(&d, (size_t)blocks * threads * 4));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_sad_u8", k_sad_u8, 1}, {"v_sad_hi_u8", k_sad_hi_u8, 1},
    {"v_sad_u16", k_sad_u16, 1}, {"v_sad_u32", k_sad_u32, 1}, {"v_msad_u8", k_msad_u8, 1},
    {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_pk_sub_u16", k_pk_sub_u16, 1}, {"v_pk_min_u16", k_pk_min_u16, 1},
    {"v_pk_max_u16", k_pk_max_u16, 1}, {"v_pk_sub_i16", k_pk_sub_i16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1},
    {"v_min_u32", k_min_u32, 1}, {"v_min3_u32", k_min3_u32, 1}, {"v_max3_u32", k_max3_u32, 1}, {"v_med3_u32", k_med3_u32, 1},
    {"v_perm_b32", k_perm_b32, 1}, {"v_alignbyte_b32", k_alignbyte, 1}, {"v_bfe_u32", k_bfe_u32, 1},
    {"v_lshl_or_b32", k_lshl_or, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_add3_u32", k_add3_u32, 1},
    {"v_lshl_add_u32", k_lshl_add, 1}, {"v_xad_u32", k_xad_u32, 1}, {"v_cndmask_b32", k_cndmask, 1},
    {"v_add_u32_dpp row_shr:1", k_add_dpp_rshr1, 1}, {"v_add_u32_dpp row_shr:4", k_add_dpp_rshr4, 1},
    {"v_add_u32_dpp wave_shr:1", k_add_dpp_wshr1, 1}, {"v_add_u32_dpp wave_shl:1", k_add_dpp_wshl1, 1},
    {"v_add_u32_dpp row_bcast:15", k_add_dpp_bc15, 1}, {"v_mov_b32_dpp row_shr:1", k_mov_dpp_rshr1, 1},
    {"v_sub_u32_sdwa BYTE_1", k_sub_sdwa, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
    {"and_b32", k_and_b32, 1}, {"or_b32", k_or_b32, 1}, {"xor_b32", k_xor_b32, 1}, {"lshlrev_b32", k_lshlrev_b32, 1}, {"lshrrev_b32", k_lshrrev_b32, 1}, {"max_u32", k_max_u32, 1}, {"mov_b32", k_mov_b32, 1}, {"add_u16", k_add_u16, 1}, {"min_u16", k_min_u16, 1}, {"mul_u32_u24", k_mul_u32_u24, 1}, {"add_f32", k_add_f32, 1}, {"fma_f32", k_fma_f32, 1}, {"add_u32_e64", k_add_u32_e64, 1}, {"add_co_u32", k_add_co_u32, 1}, {"add_u32_sdwa", k_add_u32_sdwa, 1}, {"add_lshl", k_add_lshl, 1}, {"cmp_cnd", k_cmp_cnd, 2}, {"cnd_e64", k_cnd_e64, 1}, {"cmp_only", k_cmp_only, 2}, {"mix_add_sad", k_mix_add_sad, 2}, {"mix_add_add_sad", k_mix_add_add_sad, 3}, {"pk_lshl_b16", k_pk_lshl_b16, 1}, {"pk_max_i16", k_pk_max_i16, 1}, {"sub_u32_sgpr", k_sub_u32_sgpr, 1}, {"sad_u8_sgpr", k_sad_u8_sgpr, 1}, {"addc_u32", k_addc_u32, 1}, 
    {"[24 chains] v_add_u32", kw_add_u32, 1}, {"[24 chains] v_sad_u32", kw_sad_u32, 1}, {"[24 chains] v_sad_u8", kw_sad_u8, 1},
    {"[24 chains] v_min3_u32", kw_min3_u32, 1}, {"[24 chains] v_min_u32", kw_min_u32, 1}, {"[24 chains] v_add_u32_dpp wshr", kw_add_dpp, 1},
    {"[24 chains] v_pk_add_u16", kw_pk_add_u16, 1}, {"[24 chains] v_add3_u32", kw_add3_u32, 1}, {"[24 chains] v_cndmask_e64", kw_cndmask, 1},
    {"[24 chains] v_sad_u32 ..,0", kw_sad_const, 1}, {"[24 chains] v_sub_u32 sgpr", kw_sub_sgpr, 1},
    {"v_qsad_pk_u16_u8", k_qsad_pk, 1}, {"v_mqsad_pk_u16_u8", k_mqsad_pk, 1}, {"v_lshlrev_b64", k_lshl_b64, 1},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-28s %12s %14s %10s\n", "instruction", "ms", "Tlane-inst/s", "cyc/wave-inst/SIMD@nominal");
  for (auto& e : es) {
    double inst_per_lane = (std::string(e.name).rfind("[24", 0) == 0 ? (double)(ITER * 16 / 24) * 24.0 : (double)ITER * 16.0) * e.ops_per_inst;
    e.fn<<<blocks, threads>>>(d, 0x01020304u, 0x0a0b0c0du);   // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0));
      e.fn<<<blocks, threads>>>(d, 0x01020304u, 0x0a0b0c0du);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double lane_inst = inst_per_lane * blocks * threads;
    double rate = lane_inst / (best * 1e-3);
    double wave_inst_per_simd = lane_inst / 64.0 / (cus * 4.0);
    double cyc = (best * 1e-3) * clk / wave_inst_per_simd;
    printf("%-28s %12.4f %14.3f %10.2f\n", e.name, best, rate / 1e12, cyc);
  }
  // semantics
  unsigned long long* so; CK(hipMalloc(&so, 128 * 8)); CK(hipMemset(so, 0, 128 * 8));
  k_semantics<<<1, 64>>>(so); CK(hipDeviceSynchronize());
  unsigned long long h[72]; CK(hipMemcpy(h, so, sizeof(h), hipMemcpyDeviceToHost));
  printf("qsad acc overflow (65520+1020 per lane): %016llx  (wrap -> 03ec.., saturate -> ffff..)\n", h[0]);
  printf("qsad order: %016llx  expect lanes i=0..3: sum|S0[i+k]-S1[k]|\n", h[1]);
  printf("sad_u8 0x0001FFFF + 1020 = %08llx ; sad_hi_u8 0xFFFF0005 + (1020<<16) = %08llx\n", h[2], h[3]);
  printf("msad S0 zero byte: %llu  S1 zero byte: %llu (unmasked would be 16 / 4+4+5+4=17)\n", h[4], h[5]);
  printf("mqsad_pk: %016llx  with zero ref byte: %016llx\n", h[6], h[7]);
  for (int l : {0, 1, 15, 16, 17, 31, 32, 63})
    printf("lane %2d: wave_shr1=%llu wave_shl1=%llu row_shr1=%llu row_shr1(bc)=%llu\n", l,
           (h[8 + l] >> 48) & 0xffff, (h[8 + l] >> 32) & 0xffff, (h[8 + l] >> 16) & 0xffff, h[8 + l] & 0xffff);
  return 0;
}
