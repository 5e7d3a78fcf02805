// block_bench.hip -- times the two register-resident blocks of the column-wave kernel in
// isolation (no LDS/global traffic in the loop) to separate VALU issue cost from everything else.
//   hipcc --offload-arch=gfx950 -O3 -I../../steroscopic-hardware_amd/csrc -o block_bench block_bench.hip
#include "../../steroscopic-hardware_amd/csrc/colwave_kernel.h"      // device code only: Cw<...> and its blocks
#include <cstdio>
using namespace sadhip;

template <int P, int ND, int H, int MODE, bool PK, int BARK = 1>
__global__ void __launch_bounds__(1024) k_block(unsigned* io, int iters)
{
  using K = Cw<P, ND, H, PK>;
  constexpr int NC = K::NC;
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  unsigned C[P][NC]; typename K::Masks M; unsigned best[P];
  for (int k = 0; k < K::NMB; ++k) M.mb[k] = ~0ull; M.carry = 0u;
  for (int p = 0; p < P; ++p) { best[p] = 0xffffffffu; for (int q = 0; q < NC; ++q) C[p][q] = io[(threadIdx.x * P + p) * NC + q] & 0xff00ff; }
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) smem[i] = io[i] & 0xff;
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned* l = smem + 4 * (threadIdx.x & 63);
  const unsigned* r = smem + 256 + 12 * wv + 4 * (threadIdx.x & 63);
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    const int so = (MODE & 128) ? slot * 584 : 0;                    // ring rows of 584 dwords
    const int so2 = (MODE & 128) ? ((slot + 2) % 11) * 584 : 256;
    using S = typename K::Src;
    if (MODE & 1) K::update_columns(C, S{l + so, l + so}, S{r + so, r + so}, S{l + so2, l + so2}, S{r + so2, r + so2}, true);
    slot = slot + 1 == 11 ? 0 : slot + 1;
    if (MODE & 32) { for (int p = 0; p < P; ++p) for (int q = 0; q < NC; ++q) C[p][q] += (unsigned)it & 1u; }   // defeat hoisting
    if (MODE & 2) K::template reduce_row<0, 0>(C, M, best, 0u, ND);
    if (MODE & 2 || MODE & 4) { for (int p = 0; p < P; ++p) best[p] |= 0x80000000u; }   // keep the min chain from saturating
    if (MODE & 256) { unsigned* rb = smem + 7000 + (it & 1) * 256 + 4 * (threadIdx.x & 63); for (int p = 0; p < P; ++p) atomicMin(rb + p, best[p]); }
    if ((MODE & 64) && (it % BARK) == BARK - 1) __syncthreads();
    C[0][0] += best[0] & 1;      // keep a dependency so nothing is hoisted
  }
  unsigned acc = 0;
  for (int p = 0; p < P; ++p) { acc ^= best[p]; for (int q = 0; q < NC; ++q) acc ^= C[p][q]; }
  io[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int P, int ND, int H, int MODE, bool PK, int BARK = 1>
void run(const char* name, int nw, int wgs_per_cu)
{
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * wgs_per_cu, iters = 2000;
  unsigned* d; hipMalloc(&d, (size_t)blocks * 1024 * 64 * 4); hipMemset(d, 1, (size_t)blocks * 1024 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_block<P, ND, H, MODE, PK, BARK><<<blocks, 64 * nw, 32768>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_block<P, ND, H, MODE, PK, BARK><<<blocks, 64 * nw, 32768>>>(d, iters);
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
  // one workgroup per CU (config 5, BlockSize >= 16): does the per-row barrier cost issue slots?
  run<4, 12, 4, 3, true>("pk ND12 upd+red", 16, 1);
  run<4, 12, 4, 3 + 64, true>("pk ND12 upd+red +barrier/1", 16, 1);
  run<4, 12, 4, 3 + 64, true, 2>("pk ND12 upd+red +barrier/2", 16, 1);
  run<4, 12, 4, 3 + 64, true, 4>("pk ND12 upd+red +barrier/4", 16, 1);
  run<4, 12, 4, 3 + 64 + 128 + 256, true>("pk ND12 upd+red +all", 16, 1);
  run<4, 12, 4, 3 + 64 + 128 + 256, true, 2>("pk ND12 upd+red +all, barrier/2", 16, 1);
  run<4, 12, 4, 3, true>("pk ND12 upd+red", 8, 1);
  run<4, 12, 4, 3 + 64, true>("pk ND12 upd+red +barrier/1", 8, 1);
  run<4, 12, 4, 3 + 64, true, 2>("pk ND12 upd+red +barrier/2", 8, 1);
  run<4, 12, 4, 3 + 64, true>("pk ND12 upd+red +barrier/1", 8, 2);
  run<4, 12, 4, 3 + 64, true>("pk ND12 upd+red +barrier/1", 4, 5);
  return 0;
}
