// pcie_bw.hip -- what the bus gives page-locked memory on this box: H2D alone, D2H alone, both at once on two
// streams, per copy size; plus H2D split over two streams.  Sets the ceiling of the host-fed numbers in bench.py
// (a 1080p pair is 4.15 MB in, 2.07 MB out).   usage: pcie_bw [device]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
int main(int argc, char** argv)
{
  const int dev = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipSetDevice(dev));
  const size_t total = (size_t)512 << 20;
  char *h_in = nullptr, *h_out = nullptr, *d_a = nullptr, *d_b = nullptr;
  CK(hipHostMalloc((void**)&h_in, total, hipHostMallocPortable)); CK(hipHostMalloc((void**)&h_out, total, hipHostMallocPortable));
  CK(hipMalloc((void**)&d_a, total)); CK(hipMalloc((void**)&d_b, total));
  memset(h_in, 1, total); memset(h_out, 2, total);
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  printf("%10s %12s %12s %14s %14s %16s\n", "copy", "H2D GB/s", "D2H GB/s", "both: H2D", "both: D2H", "H2D on 2 streams");
  for (size_t sz : {(size_t)2 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
    const int n = (int)(total / sz);
    auto run = [&](int mode) -> double {            // 0 h2d, 1 d2h, 2 both, 3 h2d on two streams
      (void)hipDeviceSynchronize();
      const auto t0 = clk::now();
      for (int i = 0; i < n; ++i) {
        if (mode == 0 || mode == 2) (void)hipMemcpyAsync(d_a + i * sz, h_in + i * sz, sz, hipMemcpyHostToDevice, s0);
        if (mode == 1 || mode == 2) (void)hipMemcpyAsync(h_out + i * sz, d_b + i * sz, sz, hipMemcpyDeviceToHost, s1);
        if (mode == 3) (void)hipMemcpyAsync(d_a + i * sz, h_in + i * sz, sz, hipMemcpyHostToDevice, (i & 1) ? s1 : s0);
      }
      (void)hipDeviceSynchronize();
      return std::chrono::duration<double>(clk::now() - t0).count();
    };
    run(0); run(1);
    const double a = run(0), b = run(1), c = run(2), d = run(3);
    printf("%7zu MB %12.1f %12.1f %14.1f %14.1f %16.1f\n", sz >> 20, total / a / 1e9, total / b / 1e9, total / c / 1e9, total / c / 1e9, total / d / 1e9);
  }
  return 0;
}
