// valu_rates2.hip -- round-2 companion of valu_rates.hip: issue rates of the f32 / u16 VALU
// instructions a float-accumulator SAD kernel would be built from, LDS cross-lane transfers
// (ds_bpermute_b32) beside VALU work, the cost of a wave-wide argmin (the "lane = candidate"
// layout), and whether v_min_f32 / v_min3_f32 order non-negative u32 bit patterns like v_min_u32
// (denormal range included).  Diagnostic tool only; numbers are recorded in DESIGN.md.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip && ./valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 2048;

// %0 = accumulator (in/out), %1 %2 = inputs.  Floats: a = lane + 1, b = 3*lane + 0.5 (finite, normal).
#define KERNEL32(NAME, ASM)                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned va = a + (threadIdx.x << 13), vb = b + threadIdx.x * 0x3000u;         \
  unsigned c0 = a + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3,          \
           c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;                   \
  for (int i = 0; i < ITER; ++i) {                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                              \
      asm volatile(ASM : "+v"(c0) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c1) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c2) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c3) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c4) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c5) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c6) : "v"(va), "v"(vb) : "vcc");                   \
      asm volatile(ASM : "+v"(c7) : "v"(va), "v"(vb) : "vcc");                   \
    }                                                                            \
  }                                                                              \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; \
}

// two-instruction statements with a temporary (%3): dependent pair, as in  t = l - r ; c += |t|
#define KERNEL32T(NAME, ASM)                                                     \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned va = a + (threadIdx.x << 13), vb = b + threadIdx.x * 0x3000u;         \
  unsigned c0 = a + threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3,          \
           c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;                   \
  unsigned t0, t1, t2, t3, t4, t5, t6, t7;                                       \
  for (int i = 0; i < ITER; ++i) {                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                              \
      asm volatile(ASM : "+v"(c0), "=&v"(t0) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c1), "=&v"(t1) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c2), "=&v"(t2) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c3), "=&v"(t3) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c4), "=&v"(t4) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c5), "=&v"(t5) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c6), "=&v"(t6) : "v"(va), "v"(vb) : "vcc");        \
      asm volatile(ASM : "+v"(c7), "=&v"(t7) : "v"(va), "v"(vb) : "vcc");        \
    }                                                                            \
  }                                                                              \
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7; \
}

// 64-bit accumulators (v_pk_*_f32)
#define KERNEL64(NAME, ASM)                                                      \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned a, unsigned b) { \
  unsigned long long va = ((unsigned long long)(a + (threadIdx.x << 13)) << 32) | (a + threadIdx.x); \
  unsigned long long vb = ((unsigned long long)(b + threadIdx.x * 0x3000u) << 32) | b; \
  unsigned long long c0 = va + 1, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3,         \
           c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;                   \
  for (int i = 0; i < ITER; ++i) {                                               \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                              \
      asm volatile(ASM : "+v"(c0) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c1) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c2) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c3) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c4) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c5) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c6) : "v"(va), "v"(vb));                           \
      asm volatile(ASM : "+v"(c7) : "v"(va), "v"(vb));                           \
    }                                                                            \
  }                                                                              \
  unsigned long long r = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r ^ (unsigned)(r >> 32);\
}

KERNEL32(k_add_u32,       "v_add_u32 %0, %1, %0")
KERNEL32(k_sad_u32,       "v_sad_u32 %0, %1, %2, %0")
KERNEL32(k_sub_f32,       "v_sub_f32 %0, %0, %1")
KERNEL32(k_add_f32,       "v_add_f32 %0, %1, %0")
KERNEL32(k_add_f32_abs,   "v_add_f32_e64 %0, %0, |%1|")
KERNEL32(k_sub_f32_abs,   "v_sub_f32_e64 %0, %0, |%1|")
KERNEL32(k_mul_f32,       "v_mul_f32 %0, %1, %0")
KERNEL32(k_fmac_f32,      "v_fmac_f32 %0, %1, %2")
KERNEL32(k_fma_f32,       "v_fma_f32 %0, %1, %2, %0")
KERNEL32(k_fma_f32_abs,   "v_fma_f32 %0, |%1|, %2, %0")
KERNEL32(k_min_f32,       "v_min_f32 %0, %1, %0")
KERNEL32(k_max_f32,       "v_max_f32 %0, %1, %0")
KERNEL32(k_min3_f32,      "v_min3_f32 %0, %1, %2, %0")
KERNEL32(k_max3_f32,      "v_max3_f32 %0, %1, %2, %0")
KERNEL32(k_med3_f32,      "v_med3_f32 %0, %1, %2, %0")
KERNEL32(k_min_u32,       "v_min_u32 %0, %1, %0")
KERNEL32(k_min3_u32,      "v_min3_u32 %0, %1, %2, %0")
KERNEL32(k_min_i32,       "v_min_i32 %0, %1, %0")
KERNEL32(k_max_i32,       "v_max_i32 %0, %1, %0")
KERNEL32(k_sub_u16,       "v_sub_u16 %0, %0, %1")
KERNEL32(k_max_u16,       "v_max_u16 %0, %1, %0")
KERNEL32(k_min_u16,       "v_min_u16 %0, %1, %0")
KERNEL32(k_min_i16,       "v_min_i16 %0, %1, %0")
KERNEL32(k_lshr16,        "v_lshrrev_b32 %0, 16, %0")
KERNEL32(k_ashr,          "v_ashrrev_i32 %0, 1, %0")
KERNEL32(k_subrev_u32,    "v_subrev_u32 %0, %1, %0")
KERNEL32(k_cvt_ub0,       "v_cvt_f32_ubyte0 %0, %0")
KERNEL32(k_cvt_ub1,       "v_cvt_f32_ubyte1 %0, %0")
KERNEL32(k_cvt_ub3,       "v_cvt_f32_ubyte3 %0, %0")
KERNEL32(k_cvt_f32_u32,   "v_cvt_f32_u32 %0, %0")
KERNEL32(k_cvt_u32_f32,   "v_cvt_u32_f32 %0, %0")
KERNEL32(k_add_f16,       "v_add_f16 %0, %1, %0")
KERNEL32(k_pk_add_f16,    "v_pk_add_f16 %0, %1, %0")
KERNEL32(k_pk_min_f16,    "v_pk_min_f16 %0, %1, %0")
KERNEL32(k_pk_min_u16,    "v_pk_min_u16 %0, %1, %0")
KERNEL32(k_pk_add_u16,    "v_pk_add_u16 %0, %1, %0")
KERNEL32(k_bfi,           "v_bfi_b32 %0, %1, %2, %0")
KERNEL32(k_and_or,        "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_perm,          "v_perm_b32 %0, %1, %0, %2")
KERNEL32(k_pack_b32_f16,  "v_pack_b32_f16 %0, %1, %0")
KERNEL32(k_add_f32_dpp,   "v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_min_f32_dpp,   "v_min_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_mad_mix,       "v_fma_mix_f32 %0, %1, %2, %0")
KERNEL32(k_dot2_u16,      "v_dot2_u32_u16 %0, %1, %2, %0")
KERNEL32T(k_sub_addabs,   "v_sub_f32 %1, %2, %3\n\tv_add_f32_e64 %0, %0, |%1|")
KERNEL32T(k_pkmin_add,    "v_pk_min_u16 %1, %2, %3\n\tv_add_u32 %0, %0, %1")
KERNEL32T(k_minu16_add,   "v_min_u16 %1, %2, %3\n\tv_add_u32 %0, %0, %1")
KERNEL32T(k_minf32_add,   "v_min_f32 %1, %2, %3\n\tv_add_f32 %0, %0, %1")
KERNEL64(k_pk_add_f32,    "v_pk_add_f32 %0, %1, %0")
KERNEL64(k_pk_fma_f32,    "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL64(k_pk_mul_f32,    "v_pk_mul_f32 %0, %1, %0")

// ---- ds_bpermute_b32 beside VALU adds: NB transfers and NV float adds per iteration -------------
template <int NB, int NV>
__global__ void __launch_bounds__(256) k_bperm(unsigned* out, unsigned a, unsigned b)
{
  const int lane = threadIdx.x & 63;
  const int left = ((lane + 63) & 63) * 4;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = (float)(threadIdx.x + k);
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (float)(a + k + lane);
  const float inc = __uint_as_float(b);
  for (int i = 0; i < ITER / 4; ++i) {
    float r[8];
#pragma unroll
    for (int k = 0; k < NB; ++k) r[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(left, __float_as_int(x[k])));
    // independent VALU work issued while the transfers are in flight
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[v & 7]) : "v"(inc));
#pragma unroll
    for (int k = 0; k < NB; ++k) x[k] += r[k] * 0.5f;          // consumes the transfers: NB v_fmac
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k] + x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
}

// same shape with DPP adds instead of the LDS crossbar
template <int NB, int NV>
__global__ void __launch_bounds__(256) k_dppadd(unsigned* out, unsigned a, unsigned b)
{
  const int lane = threadIdx.x & 63;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = (float)(threadIdx.x + k);
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = (float)(a + k + lane);
  const float inc = __uint_as_float(b);
  for (int i = 0; i < ITER / 4; ++i) {
#pragma unroll
    for (int k = 0; k < NB; ++k)
      asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[k]) : "v"(acc[k]));
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[v & 7]) : "v"(inc));
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc[k] + x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
}

// ---- wave-wide argmin of one u32 key per lane (lane = candidate layout): 6 DPP steps per pixel ----
__global__ void __launch_bounds__(256) k_wave_argmin(unsigned* out, unsigned a, unsigned b)
{
  unsigned k[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) k[q] = (a * (threadIdx.x + 1) + q * b) | 1u;
  unsigned acc = 0;
  for (int i = 0; i < ITER / 4; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      unsigned v = k[q] + i;
      // row_shr 1,2,4(3),8 then row_bcast 15 / 31: the classic 6-step wave reduction
      asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                   "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
      acc ^= v;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- does v_min_f32 / v_min3_f32 order non-negative bit patterns like v_min_u32? -----------------
__global__ void k_minf_check(const unsigned* a, const unsigned* b, const unsigned* c, unsigned* bad, unsigned* mode, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { unsigned m; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(m)); mode[0] = m; }
  if (i >= n) return;
  unsigned x = a[i], y = b[i], z = c[i], r2, r3;
  asm volatile("v_min_f32 %0, %1, %2" : "=v"(r2) : "v"(x), "v"(y));
  asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(r3) : "v"(x), "v"(y), "v"(z));
  const unsigned e2 = x < y ? x : y;
  const unsigned e3 = e2 < z ? e2 : z;
  if (r2 != e2) atomicAdd(&bad[0], 1u);
  if (r3 != e3) atomicAdd(&bad[1], 1u);
}

struct Entry { const char* name; void (*fn)(unsigned*, unsigned, unsigned); double ops_per_stmt; double stmts; };

int main()
{
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount; double clk = p.clockRate * 1e3;
  printf("device: %s  CUs=%d  clockRate=%.0f MHz  arch=%s\n", p.name, cus, clk / 1e6, p.gcnArchName);
  const int blocks = cus * 16, threads = 256;
  unsigned* d; CK(hipMalloc(&d, (size_t)blocks * threads * 4));
  const unsigned FA = 0x3f800000u /* 1.0f */, FB = 0x40400000u /* 3.0f */;
  const double S16 = (double)ITER * 16.0;
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1, S16}, {"v_sad_u32", k_sad_u32, 1, S16},
    {"v_sub_f32", k_sub_f32, 1, S16}, {"v_add_f32", k_add_f32, 1, S16}, {"v_add_f32_e64 |abs|", k_add_f32_abs, 1, S16},
    {"v_sub_f32_e64 |abs|", k_sub_f32_abs, 1, S16}, {"v_mul_f32", k_mul_f32, 1, S16}, {"v_fmac_f32", k_fmac_f32, 1, S16},
    {"v_fma_f32", k_fma_f32, 1, S16}, {"v_fma_f32 |abs|", k_fma_f32_abs, 1, S16},
    {"v_min_f32", k_min_f32, 1, S16}, {"v_max_f32", k_max_f32, 1, S16}, {"v_min3_f32", k_min3_f32, 1, S16},
    {"v_max3_f32", k_max3_f32, 1, S16}, {"v_med3_f32", k_med3_f32, 1, S16},
    {"v_min_u32", k_min_u32, 1, S16}, {"v_min3_u32", k_min3_u32, 1, S16}, {"v_min_i32", k_min_i32, 1, S16}, {"v_max_i32", k_max_i32, 1, S16},
    {"v_sub_u16", k_sub_u16, 1, S16}, {"v_max_u16", k_max_u16, 1, S16}, {"v_min_u16", k_min_u16, 1, S16}, {"v_min_i16", k_min_i16, 1, S16},
    {"v_lshrrev_b32 16", k_lshr16, 1, S16}, {"v_ashrrev_i32", k_ashr, 1, S16}, {"v_subrev_u32", k_subrev_u32, 1, S16},
    {"v_cvt_f32_ubyte0", k_cvt_ub0, 1, S16}, {"v_cvt_f32_ubyte1", k_cvt_ub1, 1, S16}, {"v_cvt_f32_ubyte3", k_cvt_ub3, 1, S16},
    {"v_cvt_f32_u32", k_cvt_f32_u32, 1, S16}, {"v_cvt_u32_f32", k_cvt_u32_f32, 1, S16},
    {"v_add_f16", k_add_f16, 1, S16}, {"v_pk_add_f16", k_pk_add_f16, 1, S16}, {"v_pk_min_f16", k_pk_min_f16, 1, S16},
    {"v_pk_min_u16", k_pk_min_u16, 1, S16}, {"v_pk_add_u16", k_pk_add_u16, 1, S16},
    {"v_bfi_b32", k_bfi, 1, S16}, {"v_and_or_b32", k_and_or, 1, S16}, {"v_perm_b32", k_perm, 1, S16}, {"v_pack_b32_f16", k_pack_b32_f16, 1, S16},
    {"v_add_f32_dpp wave_shr:1", k_add_f32_dpp, 1, S16}, {"v_min_f32_dpp row_shr:1", k_min_f32_dpp, 1, S16},
    {"v_fma_mix_f32", k_mad_mix, 1, S16}, {"v_dot2_u32_u16", k_dot2_u16, 1, S16},
    {"pair: v_sub_f32 ; v_add_f32 |t|", k_sub_addabs, 2, S16}, {"pair: v_pk_min_u16 ; v_add_u32", k_pkmin_add, 2, S16},
    {"pair: v_min_u16 ; v_add_u32", k_minu16_add, 2, S16}, {"pair: v_min_f32 ; v_add_f32", k_minf32_add, 2, S16},
    {"v_pk_add_f32 (2 adds)", k_pk_add_f32, 1, S16}, {"v_pk_fma_f32 (2 fmas)", k_pk_fma_f32, 1, S16}, {"v_pk_mul_f32", k_pk_mul_f32, 1, S16},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](void (*fn)(unsigned*, unsigned, unsigned)) {
    fn<<<blocks, threads>>>(d, FA, FB); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0)); fn<<<blocks, threads>>>(d, FA, FB); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return (double)best;
  };
  const double waves_per_simd_total = (double)blocks * threads / 64.0 / (cus * 4.0);
  printf("%-36s %10s %28s\n", "instruction", "ms", "cyc per wave-instruction per SIMD @nominal");
  for (auto& e : es) {
    const double ms = timeit(e.fn);
    const double winst = e.stmts * e.ops_per_stmt * waves_per_simd_total;
    printf("%-36s %10.4f %12.2f\n", e.name, ms, ms * 1e-3 * clk / winst);
  }
  // LDS crossbar transfers beside VALU work: cycles per ITERATION per wave per SIMD
  struct B { const char* name; void (*fn)(unsigned*, unsigned, unsigned); int nb, nv; };
  std::vector<B> bs = {
    {"bpermute x8 + 0 v_add", k_bperm<8, 0>, 8, 0}, {"bpermute x8 + 16 v_add", k_bperm<8, 16>, 8, 16},
    {"bpermute x8 + 32 v_add", k_bperm<8, 32>, 8, 32}, {"bpermute x8 + 56 v_add", k_bperm<8, 56>, 8, 56},
    {"bpermute x4 + 56 v_add", k_bperm<4, 56>, 4, 56}, {"bpermute x0 + 56 v_add", k_bperm<0, 56>, 0, 56},
    {"dpp-add  x8 + 0 v_add", k_dppadd<8, 0>, 8, 0}, {"dpp-add  x8 + 56 v_add", k_dppadd<8, 56>, 8, 56},
    {"dpp-add  x0 + 56 v_add", k_dppadd<0, 56>, 0, 56},
  };
  printf("\n%-28s %10s %24s\n", "transfer mix", "ms", "cyc per iteration per wave-slot (SIMD-cycles / iteration)");
  for (auto& b : bs) {
    const double ms = timeit(b.fn);
    printf("%-28s %10.4f %12.1f\n", b.name, ms, ms * 1e-3 * clk / ((double)(ITER / 4) * waves_per_simd_total));
  }
  {
    const double ms = timeit(k_wave_argmin);
    printf("\nwave argmin (6 DPP min steps per register): %.4f ms = %.1f SIMD-cycles per reduced register (64 candidates of one pixel)\n",
           ms, ms * 1e-3 * clk / ((double)(ITER / 4) * 8.0 * waves_per_simd_total));
  }
  // v_min_f32 as an integer minimum
  {
    const int n = 1 << 20;
    std::vector<unsigned> a(n), b(n), c(n);
    unsigned long long st = 12345;
    auto rnd = [&]() { st += 0x9E3779B97F4A7C15ull; unsigned long long z = st; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return (unsigned)((z ^ (z >> 31)) >> 16); };
    const unsigned lim[4] = {1u << 23, 1u << 26, 0x7f800000u, 1u << 10};
    const char* nm[4] = {"[0,2^23) (all denormal)", "[0,2^26)", "[0,0x7f800000)", "[0,1024) close values"};
    unsigned *da, *db, *dc, *dbad, *dmode;
    CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&dbad, 8)); CK(hipMalloc(&dmode, 4));
    for (int t = 0; t < 4; ++t) {
      for (int i = 0; i < n; ++i) { a[i] = rnd() % lim[t]; b[i] = rnd() % lim[t]; c[i] = rnd() % lim[t]; if ((i & 7) == 0) b[i] = a[i]; }
      CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(dbad, 0, 8));
      k_minf_check<<<n / 256, 256>>>(da, db, dc, dbad, dmode, n); CK(hipDeviceSynchronize());
      unsigned bad[2], mode; CK(hipMemcpy(bad, dbad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&mode, dmode, 4, hipMemcpyDeviceToHost));
      printf("v_min_f32 vs umin on %-28s: %u / %d mismatches; v_min3_f32: %u   (MODE=0x%08x: fp_denorm=%u ieee=%u dx10_clamp=%u)\n",
             nm[t], bad[0], n, bad[1], mode, (mode >> 4) & 0xf, (mode >> 9) & 1, (mode >> 8) & 1);
    }
  }
  return 0;
}
