// valu_mix.hip -- does a fast-class VALU instruction cost issue time next to slow-class ones?
// Loop bodies of S slow (v_sad_u8) and F fast (v_add_u32) instructions on independent registers,
// interleaved, at 5 waves per SIMD (the occupancy of the BASELINE config-3 kernel).  Prints
// SIMD-cycles per loop iteration per wave next to the additive model S*4.25 + F*2.3.
//   hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096;

template <int S, int F, int NOP>
__global__ void __launch_bounds__(256) k_mix(unsigned* out, unsigned a, unsigned b)
{
  unsigned va = a ^ threadIdx.x, vb = b + threadIdx.x * 3u;
  unsigned s[16], f[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { s[k] = threadIdx.x + k; f[k] = threadIdx.x * 5 + k; }
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < S) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(s[k]) : "v"(va), "v"(vb));
      if (k < F) asm volatile("v_add_u32 %0, %1, %0" : "+v"(f[k]) : "v"(va));
      if (k < NOP) asm volatile("s_nop 0");
    }
  }
  unsigned r = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) r ^= s[k] ^ f[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int S, int F, int NOP>
void run(int waves_per_simd)
{
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
  unsigned* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  // pad LDS so exactly waves_per_simd blocks are resident per CU: all blocks run in one round
  const size_t lds = 160 * 1024 / waves_per_simd - 1024;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<S, F, NOP>), dim3(blocks), dim3(256), lds > 64000 ? 64000 : lds, 0, d, 1u, 2u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<S, F, NOP>), dim3(blocks), dim3(256), lds > 64000 ? 64000 : lds, 0, d, 1u, 2u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double cyc = best * 1e-3 * p.clockRate * 1e3 / ITER / waves_per_simd;
  printf("S=%2d F=%2d nop=%2d  %d waves/SIMD: %7.1f SIMD-cycles per iteration per wave   additive model %.1f   slow-only %.1f\n",
         S, F, NOP, waves_per_simd, cyc, S * 4.25 + F * 2.3 + NOP * 4.0, S * 4.25);
  CK(hipFree(d));
}

int main()
{
  for (int w : {5, 2, 1}) {
    run<8, 0, 0>(w); run<8, 2, 0>(w); run<8, 4, 0>(w); run<8, 8, 0>(w); run<8, 16, 0>(w);
    run<0, 8, 0>(w); run<4, 8, 0>(w); run<8, 0, 4>(w); run<8, 4, 4>(w); run<16, 4, 0>(w);
  }
  return 0;
}
