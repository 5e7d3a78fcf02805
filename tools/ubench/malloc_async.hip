// malloc_async.hip -- what a stream-ordered allocation of the multi-pass key scratch (<= 256 MB, launch_fast in
// kernels_fast.hip) costs per call: hipMallocAsync + a kernel + hipFreeAsync + stream synchronise, with the default memory
// pool's release threshold at its default (0: unused memory goes back to the driver at every synchronise) and raised
// (memory stays in the pool).  Round 4: on some boxes of the pool a multi-pass job (MaxDisparity 192 / 256 with BlockSize >= 16)
// ran at 152 us per pair instead of 105 -- the same binary, one allocation per call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void touch(unsigned* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
static double cycle(hipStream_t s, size_t bytes, int reps)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) {
    unsigned* p = nullptr;
    if (hipMallocAsync((void**)&p, bytes, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, p, bytes / 4);
    (void)hipFreeAsync(p, s);
    (void)hipStreamSynchronize(s);
  }
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main()
{
  hipStream_t s; if (hipStreamCreate(&s) != hipSuccess) return 1;
  hipMemPool_t pool; if (hipDeviceGetDefaultMemPool(&pool, 0) != hipSuccess) return 1;
  unsigned long long thr = 0;
  (void)hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  printf("default release threshold of the device's memory pool: %llu bytes\n", thr);
  for (size_t mb : {8, 64, 256}) {
    (void)cycle(s, mb << 20, 2);
    printf("%4zu MB, threshold %llu: %.3f ms per malloc + kernel + free + synchronise\n", mb, thr, cycle(s, mb << 20, 10));
  }
  unsigned long long big = 1ull << 30;
  if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &big) != hipSuccess) return 1;
  for (size_t mb : {8, 64, 256}) {
    (void)cycle(s, mb << 20, 2);
    printf("%4zu MB, threshold %llu: %.3f ms per malloc + kernel + free + synchronise\n", mb, big, cycle(s, mb << 20, 10));
  }
  return 0;
}
