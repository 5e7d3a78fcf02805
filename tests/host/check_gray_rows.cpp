// host/gray_rows.h (the luminance the copy threads write when SADHIP_FLAG_HOST_GRAY is set) against the scalar formula of
// pkg/despair/gray.go:43-58 on color.NRGBA.RGBA(), built under AddressSanitizer + UBSan by tests/test_copy_gang_cpu.py:
//  - the exact integer identities the vector code stands on, for all 65 536 (c8, a8);
//  - every width 1..199 for RGB and RGBA sources allocated at EXACTLY w * channels bytes (a read past a row's end is an ASan
//    error), destinations checked for bytes beyond w; opaque / near-opaque / sparse / arbitrary alpha;
//  - every (r, g) with four b values and mixed alpha;
//  - through a CopyGang (rows split over threads), strided source rows, exact-size source.
#include "../../steroscopic-hardware_amd/host/copy_gang.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace sadhip;

int main()
{
  for (unsigned c = 0; c < 256; ++c)
    for (unsigned a = 0; a < 256; ++a) {
      const unsigned t = c * a, u = 514 * t, got = t + ((u + (u >> 16) + 1) >> 16), ref = c * 0x101u * (a * 0x101u) / 0xffffu;
      if (got != ref) { printf("identity fails at c=%u a=%u\n", c, a); return 1; }
      if (a == 255 && ref != 257 * c) { printf("opaque identity fails at c=%u\n", c); return 1; }
    }
  int bad = 0;
  srand(1);
  for (int ch = 3; ch <= 4; ++ch)
    for (int w = 1; w < 200; ++w) {
      std::vector<uint8_t> src((size_t)w * ch), d1(w + 16), d2(w + 16);
      for (int mode = 0; mode < 4; ++mode) {
        for (auto& v : src) v = (uint8_t)rand();
        if (ch == 4)
          for (int x = 0; x < w; ++x) {
            if (mode == 0) src[4 * x + 3] = 255;
            else if (mode == 1) src[4 * x + 3] = (rand() % 8 == 0) ? (uint8_t)rand() : 255;
            else if (mode == 2) src[4 * x + 3] = (uint8_t)(251 + rand() % 5);
          }
        std::fill(d1.begin(), d1.end(), 0xEE); std::fill(d2.begin(), d2.end(), 0xEE);
        gray_row(d1.data(), src.data(), w, ch);
        gray_row_scalar(d2.data(), src.data(), w, ch);
        if (memcmp(d1.data(), d2.data(), w + 16)) { printf("mismatch channels %d w %d mode %d\n", ch, w, mode); ++bad; }
      }
    }
  {
    const size_t n = 256 * 256 * 4;
    std::vector<uint8_t> src(n * 4), d1(n), d2(n);
    size_t k = 0;
    for (int r = 0; r < 256; ++r) for (int g = 0; g < 256; ++g) for (int b = 0; b < 4; ++b, ++k) {
      src[k * 4] = (uint8_t)r; src[k * 4 + 1] = (uint8_t)g; src[k * 4 + 2] = (uint8_t)(b * 85); src[k * 4 + 3] = (k % 3) ? 255 : (uint8_t)(r ^ g);
    }
    gray_row(d1.data(), src.data(), (int)n, 4);
    gray_row_scalar(d2.data(), src.data(), (int)n, 4);
    if (memcmp(d1.data(), d2.data(), n)) { printf("mismatch on the (r, g, b) sweep\n"); ++bad; }
  }
  for (int helpers = 0; helpers <= 5; helpers += 5)
    for (int ch = 3; ch <= 4; ++ch) {
      CopyGang gang(helpers);
      const int w = 1021, rows = 701;
      const size_t spitch = (size_t)w * ch + 37, dpitch = 1280;
      std::vector<uint8_t> src((size_t)(rows - 1) * spitch + (size_t)w * ch), dst((size_t)rows * dpitch, 0xEE), ref((size_t)rows * dpitch, 0xEE);
      for (auto& v : src) v = (uint8_t)rand();
      gang.rows(RowCopy{dst.data(), dpitch, src.data(), spitch, (size_t)w * ch, false, ch, w}, 0, rows);
      for (int r = 0; r < rows; ++r) gray_row_scalar(ref.data() + (size_t)r * dpitch, src.data() + (size_t)r * spitch, w, ch);
      if (dst != ref) { printf("mismatch through the gang: helpers %d channels %d\n", helpers, ch); ++bad; }
    }
  printf("gray rows: bad=%d\n", bad);
  return bad != 0;
}
