// simd_map.hip -- where does the dispatcher put the waves of a workgroup?  Records HW_ID
// (SIMD, CU, SE, XCC) for every wave of a launch shaped like the column-wave kernel.
//   hipcc --offload-arch=gfx950 -O3 -o simd_map simd_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool BAR>
__global__ void k_map(unsigned* out, int spin)
{
  extern __shared__ unsigned smem[];
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // stay resident for a while so that the launch fills the chip like the real kernel
  unsigned v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
  if (BAR) __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    out[(blockIdx.x * nw + w) * 2] = hwid;
    out[(blockIdx.x * nw + w) * 2 + 1] = (xcc & 0xf) | (v == 12345u ? 0x80000000u : 0u);
  }
}

int main(int argc, char** argv)
{
  const int nw = argc > 1 ? atoi(argv[1]) : 7;
  const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
  const int lds = argc > 3 ? atoi(argv[3]) : 36000;
  unsigned* d; hipMalloc(&d, (size_t)blocks * nw * 8);
  const bool bar = argc > 4 ? atoi(argv[4]) != 0 : true;
  if (bar) k_map<true><<<blocks, 64 * nw, lds>>>(d, 200000); else k_map<false><<<blocks, 64 * nw, lds>>>(d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h((size_t)blocks * nw * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // histogram of (wave index within WG) -> SIMD
  std::vector<int> hist(nw * 4, 0);
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < nw; ++w) hist[w * 4 + ((h[(b * nw + w) * 2] >> 4) & 3)]++;
  printf("nw=%d blocks=%d lds=%d\nwave: SIMD0 SIMD1 SIMD2 SIMD3\n", nw, blocks, lds);
  for (int w = 0; w < nw; ++w) printf("%4d: %5d %5d %5d %5d\n", w, hist[w * 4], hist[w * 4 + 1], hist[w * 4 + 2], hist[w * 4 + 3]);
  // first 12 workgroups in detail
  for (int b = 0; b < 12 && b < blocks; ++b) {
    const unsigned id = h[(b * nw) * 2];
    printf("wg %2d xcc %u se %u cu %2u simds:", b, h[(b * nw) * 2 + 1] & 0xf, (id >> 13) & 7, (id >> 8) & 15);
    for (int w = 0; w < nw; ++w) printf(" %u", (h[(b * nw + w) * 2] >> 4) & 3);
    printf("\n");
  }
  // per-CU SIMD load: how uneven are the four SIMDs of one CU?
  {
    std::vector<int> load(8 * 8 * 16 * 4, 0);
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < nw; ++w) {
        const unsigned id = h[(b * nw + w) * 2], x = h[(b * nw + w) * 2 + 1] & 0xf;
        load[(((x & 7) * 8 + ((id >> 13) & 7)) * 16 + ((id >> 8) & 15)) * 4 + ((id >> 4) & 3)]++;
      }
    int cus = 0; double sum_ratio = 0; int shown = 0;
    for (int c = 0; c < 8 * 8 * 16; ++c) {
      const int* l = &load[c * 4];
      const int tot = l[0] + l[1] + l[2] + l[3];
      if (!tot) continue;
      int mx = l[0]; for (int k = 1; k < 4; ++k) if (l[k] > mx) mx = l[k];
      sum_ratio += mx / (tot / 4.0); ++cus;
      if (shown++ < 6) printf("  cu %4d: %d %d %d %d\n", c, l[0], l[1], l[2], l[3]);
    }
    printf("bar=%d: %d CUs, mean (max SIMD load / mean SIMD load) = %.3f\n", (int)bar, cus, sum_ratio / cus);
  }
  hipFree(d);
  return 0;
}
