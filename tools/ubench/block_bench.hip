// block_bench.hip -- times the two register-resident blocks of the column-wave kernel in
// isolation (no LDS/global traffic in the loop) to separate VALU issue cost from everything else.
//   hipcc --offload-arch=gfx950 -O3 -I../../steroscopic-hardware_amd/csrc -o block_bench block_bench.hip
#include "../../steroscopic-hardware_amd/csrc/kernels_fast.hip"
#include "../../steroscopic-hardware_amd/csrc/kernels_generic.hip"
#include <cstdio>
using namespace sadhip;

template <int P, int ND, int H, int MODE, bool PK>
__global__ void __launch_bounds__(1024) k_block(unsigned* io, int iters)
{
  using K = Cw<P, ND, H, PK>;
  constexpr int NC = K::NC;
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  unsigned C[P][NC]; int jlim[P]; unsigned best[P];
  for (int p = 0; p < P; ++p) { jlim[p] = 100; best[p] = 0xffffffffu; for (int q = 0; q < NC; ++q) C[p][q] = io[(threadIdx.x * P + p) * NC + q] & 0xff00ff; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = io[i] & 0xffff;
  __syncthreads();
  const uint4* l = reinterpret_cast<const uint4*>(smem + 4 * (threadIdx.x & 63));
  const uint4* r = reinterpret_cast<const uint4*>(smem + 1024 + 4 * (threadIdx.x & 63));
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) K::update_columns(C, l, r, l + 64, r + 64);
    C[0][0] += best[0] & 1;      // keep a dependency so nothing is hoisted
  }
  unsigned acc = 0;
  for (int p = 0; p < P; ++p) { acc ^= best[p]; for (int q = 0; q < NC; ++q) acc ^= C[p][q]; }
  io[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int P, int ND, int H, int MODE, bool PK>
void run(const char* name, int nw, int wgs_per_cu)
{
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * wgs_per_cu, iters = 2000;
  unsigned* d; hipMalloc(&d, (size_t)blocks * 1024 * 64 * 4); hipMemset(d, 1, (size_t)blocks * 1024 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_block<P, ND, H, MODE, PK><<<blocks, 64 * nw, 32768>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_block<P, ND, H, MODE, PK><<<blocks, 64 * nw, 32768>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = (double)nw * wgs_per_cu / 4.0;
  const double cyc_per_iter_per_wave = ms * 1e-3 * 2.4e9 / iters / waves_per_simd;
  printf("%-34s nw=%d wg/cu=%d (%.2f waves/SIMD): %.0f SIMD-cycles per wave-iteration (%.1f per cand)\n", name, nw, wgs_per_cu,
         waves_per_simd, cyc_per_iter_per_wave, cyc_per_iter_per_wave / (P * ND));
  hipFree(d);
}

int main()
{
  run<4, 24, 4, 32, true>("packed P4 ND24 perturb only", 3, 4);
  run<4, 24, 4, 34, true>("packed P4 ND24 perturb+reduce", 3, 4);
  run<4, 24, 4, 36, true>("packed P4 ND24 perturb+reduce(mask)", 3, 4);
  run<4, 24, 4, 35, true>("packed P4 ND24 perturb+update+reduce", 3, 4);
  run<4, 12, 4, 32, true>("packed P4 ND12 perturb only", 6, 4);
  run<4, 12, 4, 34, true>("packed P4 ND12 perturb+reduce", 6, 4);
  run<4, 12, 4, 32, false>("unpacked P4 ND12 perturb only", 6, 2);
  run<4, 12, 4, 34, false>("unpacked P4 ND12 perturb+reduce", 6, 2);
  return 0;
}
