// block_bench.hip -- times the two register-resident blocks of the column-wave kernel in
// isolation (no LDS/global traffic in the loop) to separate VALU issue cost from everything else.
//   hipcc --offload-arch=gfx950 -O3 -I../../steroscopic-hardware_amd/csrc -o block_bench block_bench.hip
#include "../../steroscopic-hardware_amd/csrc/kernels_fast.hip"
#include "../../steroscopic-hardware_amd/csrc/kernels_generic.hip"
#include <cstdio>
using namespace sadhip;

template <int P, int ND, int H, int MODE>
__global__ void __launch_bounds__(P == 4 ? 1024 : 768) k_block(unsigned* io, int iters)
{
  using K = Cw<P, ND, H>;
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  unsigned C[P][ND]; int jlim[P]; unsigned best[P];
  for (int p = 0; p < P; ++p) { jlim[p] = 100; best[p] = 0xffffffffu; for (int q = 0; q < ND; ++q) C[p][q] = io[(threadIdx.x * P + p) * ND + q]; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = io[i] & 0xffff;
  __syncthreads();
  const uint4* l = reinterpret_cast<const uint4*>(smem + 4 * (threadIdx.x & 63));
  const uint4* r = reinterpret_cast<const uint4*>(smem + 1024 + 4 * (threadIdx.x & 63));
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) K::update_columns(C, l, r, l + 64, r + 64);
    if (MODE & 8) {                                   // update arithmetic only: operands already in registers
      unsigned li[P], ri[P + ND];
      for (int p = 0; p < P; ++p) li[p] = best[p & 3] + p;
      for (int m = 0; m < P + ND; ++m) ri[m] = jlim[m & 3] + m;
      for (int p = 0; p < P; ++p) for (int q = 0; q < ND; ++q) sad_acc(C[p][q], li[p], ri[p + ND - 1 - q]);
      for (int p = 0; p < P; ++p) for (int q = 0; q < ND; q += 4)
        sad_sub4(C[p][q], C[p][q + 1], C[p][q + 2], C[p][q + 3], li[p], ri[p + ND - 1 - q], ri[p + ND - 2 - q], ri[p + ND - 3 - q], ri[p + ND - 4 - q]);
    }
    if (MODE & 16) {                                  // in-row sads only
      unsigned li[P], ri[P + ND];
      for (int p = 0; p < P; ++p) li[p] = best[p & 3] + p;
      for (int m = 0; m < P + ND; ++m) ri[m] = jlim[m & 3] + m;
      for (int p = 0; p < P; ++p) for (int q = 0; q < ND; ++q) sad_acc(C[p][q], li[p], ri[p + ND - 1 - q]);
      for (int p = 0; p < P; ++p) for (int q = 0; q < ND; ++q) sad_acc(C[p][q], li[p], ri[p + ND - 1 - q]);
    }
    if (MODE & 2) K::template reduce_row<false>(C, jlim, best);
    if (MODE & 4) K::template reduce_row<true>(C, jlim, best);
    C[0][0] += best[0] & 1;      // keep a dependency so nothing is hoisted
  }
  unsigned acc = 0;
  for (int p = 0; p < P; ++p) { acc ^= best[p]; for (int q = 0; q < ND; ++q) acc ^= C[p][q]; }
  io[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int P, int ND, int H, int MODE>
void run(const char* name, int nw, int wgs_per_cu)
{
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * wgs_per_cu, iters = 2000;
  unsigned* d; hipMalloc(&d, (size_t)blocks * 1024 * 64 * 4); hipMemset(d, 1, (size_t)blocks * 1024 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_block<P, ND, H, MODE><<<blocks, 64 * nw, 32768>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_block<P, ND, H, MODE><<<blocks, 64 * nw, 32768>>>(d, iters);
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
  run<4, 12, 4, 1>("P4 ND12 H4 update", 6, 2);
  run<4, 12, 4, 8>("P4 ND12 H4 update (no LDS)", 6, 2);
  run<4, 12, 4, 16>("P4 ND12 H4 96 sads (no LDS)", 6, 2);
  run<4, 12, 4, 16>("P4 ND12 H4 96 sads (no LDS)", 4, 4);
  run<4, 12, 4, 16>("P4 ND12 H4 96 sads (no LDS)", 4, 1);
  run<4, 12, 4, 1>("P4 ND12 H4 update", 4, 4);
  run<4, 12, 4, 1>("P4 ND12 H4 update", 4, 1);
  run<4, 12, 4, 2>("P4 ND12 H4 reduce", 6, 2);
  run<4, 12, 4, 4>("P4 ND12 H4 reduce(mask)", 6, 2);
  run<4, 12, 4, 3>("P4 ND12 H4 update+reduce", 6, 2);
  run<4, 12, 4, 3>("P4 ND12 H4 update+reduce", 4, 4);
  run<4, 12, 4, 3>("P4 ND12 H4 update+reduce", 4, 1);
  run<4, 8, 4, 3>("P4 ND8 H4 update+reduce", 9, 2);
  run<8, 12, 4, 3>("P8 ND12 H4 update+reduce", 6, 2);
  run<8, 12, 8, 3>("P8 ND12 H8 update+reduce", 6, 2);
  return 0;
}
