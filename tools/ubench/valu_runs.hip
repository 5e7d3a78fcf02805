// valu_runs.hip -- what does a RUN of fast-class instructions cost between runs of slow-class ones?
// Loop body = 4 x [ RS slow instructions, then RF fast ones ], independent registers or the
// max-form pair (t = v_pk_max_u16 l, w ; c += t).  Prints SIMD-cycles per instruction pair group
// next to the additive model (slow 4.2, fast 2.3).
//   hipcc --offload-arch=gfx950 -O3 -o valu_runs valu_runs.hip && ./valu_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096;

// KIND 0: v_sad_u8 runs + independent v_add_u32 runs;  1: v_pk_max_u16 runs + independent adds;
//      2: v_pk_max_u16 into temporaries, then the adds that consume them (the max-form update);
//      3: v_sad_u8 x2 per register (enter form of the plain packed update), no fast ops
template <int RS, int RF, int KIND>
__global__ void __launch_bounds__(256) k_runs(unsigned* out, unsigned a, unsigned b)
{
  unsigned va = a ^ threadIdx.x, vb = b + threadIdx.x * 3u;
  unsigned s[16], f[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { s[k] = threadIdx.x + k; f[k] = threadIdx.x * 5 + k; }
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int k = 0; k < RS; ++k) {
        if (KIND == 0 || KIND == 3) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(s[k]) : "v"(va), "v"(vb));
        if (KIND == 1) asm volatile("v_pk_max_u16 %0, %1, %0" : "+v"(s[k]) : "v"(va));
        if (KIND == 2) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(s[k]) : "v"(va), "v"(f[(k + 8) & 15]));
      }
#pragma unroll
      for (int k = 0; k < RF; ++k) {
        if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(f[k]) : "v"(s[k % (RS ? RS : 1)]));
        else asm volatile("v_add_u32 %0, %1, %0" : "+v"(f[k]) : "v"(va));
      }
    }
  }
  unsigned r = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) r ^= s[k] ^ f[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int RS, int RF, int KIND>
void run(int waves_per_simd)
{
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * waves_per_simd;
  unsigned* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  size_t lds = 160 * 1024 / waves_per_simd - 1024; if (lds > 64000) lds = 64000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_runs<RS, RF, KIND>), dim3(blocks), dim3(256), lds, 0, d, 1u, 2u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_runs<RS, RF, KIND>), dim3(blocks), dim3(256), lds, 0, d, 1u, 2u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double cyc = best * 1e-3 * p.clockRate * 1e3 / ITER / waves_per_simd / 4;
  static const char* const kn[4] = {"sad_u8 | add", "pk_max | add", "pk_max -> add (dependent)", "sad_u8 only"};
  printf("%-28s run of %2d slow + %2d fast, %d waves/SIMD: %6.1f SIMD-cycles per group   additive model %.1f\n",
         kn[KIND], RS, RF, waves_per_simd, cyc, RS * 4.2 + RF * 2.3);
  CK(hipFree(d));
}

int main()
{
  for (int w : {5, 2}) {
    run<4, 4, 0>(w); run<8, 8, 0>(w); run<4, 4, 1>(w); run<8, 8, 1>(w); run<4, 4, 2>(w); run<8, 8, 2>(w);
    run<8, 0, 3>(w); run<8, 0, 1>(w); run<0, 8, 0>(w); run<8, 4, 0>(w); run<8, 4, 2>(w); run<16, 16, 2>(w); run<16, 8, 0>(w);
  }
  return 0;
}
