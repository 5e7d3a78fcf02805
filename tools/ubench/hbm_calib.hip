// hbm_calib.hip -- known-byte-count kernels in the access shapes the SAD kernel uses, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 for wide
// streams; other widths are uncalibrated).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.
//   k_read_u8 : every lane loads ONE byte, consecutive lanes consecutive bytes (global_load_ubyte)
//   k_write_u8: every lane stores one byte
//   k_read_x4 : 16 B per lane (reference shape)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read_u8(const unsigned char* p, size_t n, unsigned* o) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0xffffffffu) o[0] = acc;
}
__global__ void k_write_u8(unsigned char* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned char)i;
}
__global__ void k_read_x4(const uint4* p, size_t n, unsigned* o) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
  if (acc == 0xffffffffu) o[0] = acc;
}
int main() {
  const size_t n = 1ull << 30;   // 1 GiB: larger than the 256 MiB Infinity Cache
  unsigned char* d; unsigned* o; hipMalloc(&d, n); hipMalloc(&o, 4); hipMemset(d, 1, n);
  hipDeviceSynchronize();
  k_read_u8<<<4096, 256>>>(d, n, o); hipDeviceSynchronize();
  k_write_u8<<<4096, 256>>>(d, n); hipDeviceSynchronize();
  k_read_x4<<<4096, 256>>>((const uint4*)d, n / 16, o); hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", n);
  return 0;
}
