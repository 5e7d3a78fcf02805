// h2d_two_streams.hip -- round 5: do two host-to-device copies of a frame's planes (2 x 2 MB from page-locked memory, what one
// synchronous sadhip_disparity_u8 call uploads) finish sooner on TWO streams than one after the other on one?  The runtime
// gives every direction its own copy engine (round 3: uploads and downloads of one stream serialise, one stream per direction
// does not); whether two uploads on two streams get two engines is what this measures.  Also: one 4 MB copy (both planes
// adjacent), and the download of the 2 MB result overlapping nothing (reference).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
  const size_t plane = 1920 * 1080;
  unsigned char *h, *d;
  CK(hipHostMalloc((void**)&h, 4 * plane, hipHostMallocDefault));
  CK(hipMalloc((void**)&d, 4 * plane));
  memset(h, 7, 4 * plane);
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  auto bench = [&](const char* name, auto&& body) {
    std::vector<double> t;
    for (int r = 0; r < 60; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      body();
      t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    printf("%-64s median %7.1f us  min %7.1f us\n", name, t[t.size() / 2], t[0]);
  };
  bench("two 2 MB uploads, one stream, synchronise", [&] {
    (void)hipMemcpyAsync(d, h, plane, hipMemcpyHostToDevice, a);
    (void)hipMemcpyAsync(d + plane, h + plane, plane, hipMemcpyHostToDevice, a);
    (void)hipStreamSynchronize(a);
  });
  bench("two 2 MB uploads, two streams, synchronise both", [&] {
    (void)hipMemcpyAsync(d, h, plane, hipMemcpyHostToDevice, a);
    (void)hipMemcpyAsync(d + plane, h + plane, plane, hipMemcpyHostToDevice, b);
    (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
  });
  bench("two 2 MB uploads, two streams, event-joined into the first", [&] {
    (void)hipMemcpyAsync(d, h, plane, hipMemcpyHostToDevice, a);
    (void)hipMemcpyAsync(d + plane, h + plane, plane, hipMemcpyHostToDevice, b);
    (void)hipEventRecord(ev, b); (void)hipStreamWaitEvent(a, ev, 0);
    (void)hipStreamSynchronize(a);
  });
  bench("one 4 MB upload", [&] {
    (void)hipMemcpyAsync(d, h, 2 * plane, hipMemcpyHostToDevice, a);
    (void)hipStreamSynchronize(a);
  });
  bench("one 2 MB upload", [&] {
    (void)hipMemcpyAsync(d, h, plane, hipMemcpyHostToDevice, a);
    (void)hipStreamSynchronize(a);
  });
  bench("one 2 MB download", [&] {
    (void)hipMemcpyAsync(h + 2 * plane, d, plane, hipMemcpyDeviceToHost, a);
    (void)hipStreamSynchronize(a);
  });
  bench("four 1 MB uploads, one stream", [&] {
    for (int k = 0; k < 4; ++k) (void)hipMemcpyAsync(d + k * plane / 2, h + k * plane / 2, plane / 2, hipMemcpyHostToDevice, a);
    (void)hipStreamSynchronize(a);
  });
  return 0;
}
