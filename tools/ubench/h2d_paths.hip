// h2d_paths.hip -- ways to move one 1920x1080 u8 frame pair up (2 x 2 MB) and one map down (2 MB) between PAGEABLE host memory
// and the device, wall time per round trip (no kernel):
//   A  hipMemcpy2DAsync straight from / to the caller's memory (what the host-buffer entry points did through round 4: the runtime
//      page-locks the caller's pages in place and keeps those locks in a small cache keyed by the page-rounded address range)
//   B  hipMemcpyAsync (linear) straight from / to the caller's memory (the runtime stages through its own page-locked buffers)
//   C  memcpy into page-locked staging the program owns + linear async copies
//   D  hipHostRegister of the caller's memory for the duration of the call + linear async copies + hipHostUnregister
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const int W = 1920, H = 1080; const size_t N = (size_t)W * H;
  unsigned char *L = (unsigned char*)malloc(N), *R = (unsigned char*)malloc(N), *O = (unsigned char*)malloc(N);
  memset(L, 1, N); memset(R, 2, N); memset(O, 0, N);
  unsigned char *dL, *dR, *dO, *sL, *sR, *sO;
  if (hipMalloc((void**)&dL, N) || hipMalloc((void**)&dR, N) || hipMalloc((void**)&dO, N)) return 1;
  if (hipHostMalloc((void**)&sL, N, 0) || hipHostMalloc((void**)&sR, N, 0) || hipHostMalloc((void**)&sO, N, 0)) return 1;
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int reps = 30;
  for (int mode = 0; mode < 4; ++mode) {
    double best = 1e9, sum = 0;
    for (int r = 0; r < reps + 3; ++r) {
      const double t0 = now();
      if (mode == 0) {
        (void)hipMemcpy2DAsync(dL, W, L, W, W, H, hipMemcpyHostToDevice, s); (void)hipMemcpy2DAsync(dR, W, R, W, W, H, hipMemcpyHostToDevice, s);
        (void)hipMemcpy2DAsync(O, W, dO, W, W, H, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
      } else if (mode == 1) {
        (void)hipMemcpyAsync(dL, L, N, hipMemcpyHostToDevice, s); (void)hipMemcpyAsync(dR, R, N, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(O, dO, N, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
      } else if (mode == 2) {
        memcpy(sL, L, N); (void)hipMemcpyAsync(dL, sL, N, hipMemcpyHostToDevice, s);
        memcpy(sR, R, N); (void)hipMemcpyAsync(dR, sR, N, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(sO, dO, N, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); memcpy(O, sO, N);
      } else {
        if (hipHostRegister(L, N, 0) || hipHostRegister(R, N, 0) || hipHostRegister(O, N, 0)) { printf("register failed\n"); return 1; }
        (void)hipMemcpyAsync(dL, L, N, hipMemcpyHostToDevice, s); (void)hipMemcpyAsync(dR, R, N, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(O, dO, N, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        (void)hipHostUnregister(L); (void)hipHostUnregister(R); (void)hipHostUnregister(O);
      }
      const double dt = now() - t0;
      if (r >= 3) { sum += dt; if (dt < best) best = dt; }
    }
    static const char* const nm[4] = {"A 2-D copies from the caller's memory", "B linear copies from the caller's memory", "C own page-locked staging + linear",
                                       "D register for the call + linear + unregister"};
    printf("%-48s mean %.3f ms  min %.3f ms per round trip (4.1 MB up, 2.1 MB down)\n", nm[mode], sum / reps, best);
  }
  return 0;
}
