// Stress of steroscopic-hardware_amd/host/copy_gang.h (the threads that stage caller memory through a context's
// page-locked planes) on the CPU, built under ThreadSanitizer and under AddressSanitizer by tests/test_copy_gang_cpu.py.
// Random row counts, widths and pitches on both sides (whole-row copies and row-by-row copies), sizes either side of
// the gang's threshold, back-to-back and spaced calls (spin phase / sleep phase of the helpers), several gangs at once,
// gangs destroyed while their helpers spin; source buffers allocated at EXACTLY the bytes a copy may read
// ((rows-1)*pitch + wbytes) and destination rows checked for bytes outside the rows.
#include "../../steroscopic-hardware_amd/host/copy_gang.h"
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>

using sadhip::CopyGang;
using sadhip::RowCopy;

static int one_gang(unsigned seed, int helpers, int calls)
{
  std::mt19937 rng(seed);
  auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
  CopyGang gang(helpers);
  int bad = 0;
  for (int it = 0; it < calls; ++it) {
    const int rows = rnd(1, 900);
    const size_t wbytes = (size_t)rnd(1, 4000);
    const bool same_pitch = rnd(0, 2) == 0;
    const size_t spitch = wbytes + (size_t)rnd(0, 2) * (size_t)rnd(0, 300);
    const size_t dpitch = same_pitch ? spitch : wbytes + (size_t)rnd(0, 300);
    const bool whole = rnd(0, 1) == 1;                       // upload direction: dst is a plane of ours (full pitch rows)
    const size_t sbytes = (size_t)(rows - 1) * spitch + wbytes;              // exactly what may be read
    const size_t dbytes = whole ? (size_t)rows * dpitch : (size_t)(rows - 1) * dpitch + wbytes;
    std::unique_ptr<uint8_t[]> src(new uint8_t[sbytes]), dst(new uint8_t[dbytes]);
    for (size_t i = 0; i < sbytes; ++i) src[i] = (uint8_t)(i * 131u + seed + (unsigned)it);
    memset(dst.get(), 0xEE, dbytes);
    gang.rows(RowCopy{dst.get(), dpitch, src.get(), spitch, wbytes, whole}, 0, rows);
    for (int r = 0; r < rows && !bad; ++r) {
      if (memcmp(dst.get() + (size_t)r * dpitch, src.get() + (size_t)r * spitch, wbytes) != 0) { bad = 1; break; }
      if (!whole && r + 1 < rows)                            // download direction: nothing outside the caller's rows
        for (size_t k = wbytes; k < dpitch; ++k)
          if (dst[(size_t)r * dpitch + k] != 0xEE) { bad = 1; break; }
    }
    if (bad) { printf("MISMATCH seed %u call %d rows %d w %zu pitches %zu %zu whole %d\n", seed, it, rows, wbytes, spitch, dpitch, (int)whole); return 1; }
    if (rnd(0, 9) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd(50, 400)));   // past the spin phase
  }
  return 0;
}

int main(int argc, char** argv)
{
  const int calls = argc > 1 ? atoi(argv[1]) : 300;
  int bad = 0;
  // one gang after the other with 0..7 helpers, then four at once (contexts of a pool), then gangs that die young
  for (int h = 0; h <= 7; ++h) bad += one_gang(100u + (unsigned)h, h, calls);
  {
    std::vector<std::thread> th;
    std::atomic<int> b{0};
    for (int g = 0; g < 4; ++g) th.emplace_back([&, g] { b += one_gang(200u + (unsigned)g, 3 + g % 2, calls); });
    for (auto& t : th) t.join();
    bad += b.load();
  }
  for (int k = 0; k < 40; ++k) {
    CopyGang gang(5);
    std::vector<uint8_t> a((size_t)1 << 20, (uint8_t)k), d((size_t)1 << 20, 0);
    gang.rows(RowCopy{d.data(), 1024, a.data(), 1024, 1024, true}, 0, 1024);
    if (d[12345] != (uint8_t)k) ++bad;
  }                                                           // destroyed while the helpers are still in their spin phase
  printf("%s gangs=%d calls=%d violations=%d\n", bad ? "FAIL" : "OK", 8 + 4 + 40, calls, bad);
  return bad ? 1 : 0;
}
