// host/gray_rows.h -- the Go luminance of pkg/despair/gray.go:43-58 / color.GrayModel.Convert (pkg/camera/output.go:145,160)
// on the HOST, for the copy threads of the staged path (copy_gang.h): SADHIP_FLAG_HOST_GRAY of sadhip_disparity_rgba_u8.
// Plain C++17, no HIP (tests/test_copy_gang_cpu.py builds it alone, against the formula below, exhaustively per pixel value
// class and under AddressSanitizer for the bounds of every row).
//
//   NRGBA.RGBA():  c16 = c8*0x101 * (a8*0x101) / 0xffff        (integer division)
//   GrayModel:     Y   = (19595*r16 + 38470*g16 + 7471*b16 + 2^15) >> 24
//
// The copy threads read every byte of a decoded picture anyway (they stage it into page-locked memory); converting while
// they do writes and uploads a quarter of the bytes (a 1080p RGBA pair: 16.6 MB -> 4.1 MB over the bus).  The device kernel
// k_gray (kernels_generic.hip) stays the default; both are bit-identical to oracle_gray_from_nrgba.
//
// Exact integer forms used by the vector code (checked for all 65 536 (c8, a8) by the test):
//   t = c8*a8;  c16 = t + ((u + (u >> 16) + 1) >> 16)  with u = 514*t          [66049 = 65535 + 514]
//   a8 == 255:  c16 = 257*c8, so Y = (257*S + 2^15) >> 24 with S = 19595 r8 + 38470 g8 + 7471 b8
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace sadhip {

inline uint8_t gray_px(unsigned r8, unsigned g8, unsigned b8, unsigned a8)
{
  const uint32_t a = a8 * 0x101u;
  const uint32_t r = r8 * 0x101u * a / 0xffffu, g = g8 * 0x101u * a / 0xffffu, b = b8 * 0x101u * a / 0xffffu;
  return (uint8_t)((19595u * r + 38470u * g + 7471u * b + (1u << 15)) >> 24);
}

inline void gray_row_scalar(uint8_t* dst, const uint8_t* src, int w, int channels)
{
  if (channels == 4) for (int x = 0; x < w; ++x, src += 4) dst[x] = gray_px(src[0], src[1], src[2], src[3]);
  else for (int x = 0; x < w; ++x, src += 3) dst[x] = gray_px(src[0], src[1], src[2], 255);
}

#if defined(__x86_64__)
// eight pixels as dwords R | G<<8 | B<<16 | A<<24, every A == 255  ->  eight luminances in the low bytes of eight dwords
__attribute__((target("avx2"))) static inline __m256i gray8_opaque(__m256i px)
{
  // 16-bit lanes r, g, b, a per pixel; vpmaddwd with (19595, 38470 - 65536, 7471, 0) gives r*19595 - g*27066 and b*7471 per
  // pixel (signed 16-bit weights), the horizontal add their sum, and g << 16 restores the 65536*g the signed weight dropped
  const __m256i wts = _mm256_setr_epi16(19595, -27066, 7471, 0, 19595, -27066, 7471, 0, 19595, -27066, 7471, 0, 19595, -27066, 7471, 0);
  const __m256i pq = _mm256_permute4x64_epi64(px, 0xD8);                     // pixels 0 1 4 5 | 2 3 6 7
  const __m256i A = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(pq)), wts);
  const __m256i B = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_extracti128_si256(pq, 1)), wts);
  __m256i S = _mm256_hadd_epi32(A, B);                                       // pixels 0 1 2 3 | 4 5 6 7
  S = _mm256_add_epi32(S, _mm256_slli_epi32(_mm256_and_si256(px, _mm256_set1_epi32(0xff00)), 8));
  const __m256i y = _mm256_add_epi32(_mm256_add_epi32(_mm256_slli_epi32(S, 8), S), _mm256_set1_epi32(1 << 15));   // 257*S + 2^15 < 2^32
  return _mm256_srli_epi32(y, 24);
}

// ... any alpha
__attribute__((target("avx2"))) static inline __m256i premul16(__m256i c8, __m256i a8)
{
  const __m256i t = _mm256_mullo_epi16(c8, a8);                              // <= 65 025: exact in the low half, high half 0
  const __m256i u = _mm256_add_epi32(_mm256_slli_epi32(t, 9), _mm256_slli_epi32(t, 1));        // 514 t
  const __m256i q = _mm256_srli_epi32(_mm256_add_epi32(_mm256_add_epi32(u, _mm256_srli_epi32(u, 16)), _mm256_set1_epi32(1)), 16);
  return _mm256_add_epi32(t, q);
}

__attribute__((target("avx2"))) static inline __m256i gray8_alpha(__m256i px)
{
  const __m256i m8 = _mm256_set1_epi32(0xff);
  const __m256i a = _mm256_srli_epi32(px, 24);
  const __m256i r = premul16(_mm256_and_si256(px, m8), a), g = premul16(_mm256_and_si256(_mm256_srli_epi32(px, 8), m8), a),
                b = premul16(_mm256_and_si256(_mm256_srli_epi32(px, 16), m8), a);
  __m256i y = _mm256_mullo_epi32(r, _mm256_set1_epi32(19595));
  y = _mm256_add_epi32(y, _mm256_mullo_epi32(g, _mm256_set1_epi32(38470)));
  y = _mm256_add_epi32(y, _mm256_mullo_epi32(b, _mm256_set1_epi32(7471)));
  return _mm256_srli_epi32(_mm256_add_epi32(y, _mm256_set1_epi32(1 << 15)), 24);
}

__attribute__((target("avx2"))) static inline void store8(uint8_t* dst, __m256i y)
{
  const __m256i w16 = _mm256_packus_epi32(y, y);                             // per 128-bit lane: 4 words (twice)
  const __m256i b8 = _mm256_packus_epi16(w16, w16);                          // per lane: 4 bytes (four times)
  const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(b8)), hi = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(b8, 1));
  memcpy(dst, &lo, 4); memcpy(dst + 4, &hi, 4);
}

__attribute__((target("avx2"))) inline void gray_row_avx2(uint8_t* dst, const uint8_t* src, int w, int channels)
{
  int x = 0;
  if (channels == 4) {
    const __m256i amask = _mm256_set1_epi32((int)0xff000000u);
    for (; x + 8 <= w; x += 8) {
      const __m256i px = _mm256_loadu_si256((const __m256i*)(src + (size_t)x * 4));
      const bool opaque = _mm256_movemask_epi8(_mm256_cmpeq_epi32(_mm256_and_si256(px, amask), amask)) == -1;
      store8(dst + x, opaque ? gray8_opaque(px) : gray8_alpha(px));
    }
    gray_row_scalar(dst + x, src + (size_t)x * 4, w - x, 4);
    return;
  }
  // RGB: two 16-byte loads (pixels x .. x+3 at byte 3x, x+4 .. x+7 at 3x + 12) expanded to R G B 255 dwords; the second load
  // reads 4 bytes past pixel x+7, so the vector loop stops while 3x + 28 <= 3w
  const __m128i sh = _mm_setr_epi8(0, 1, 2, -1, 3, 4, 5, -1, 6, 7, 8, -1, 9, 10, 11, -1);
  const __m256i opa = _mm256_set1_epi32((int)0xff000000u);
  for (; (size_t)x * 3 + 28 <= (size_t)w * 3; x += 8) {
    const __m128i lo = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(src + (size_t)x * 3)), sh);
    const __m128i hi = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(src + (size_t)x * 3 + 12)), sh);
    const __m256i px = _mm256_or_si256(_mm256_inserti128_si256(_mm256_castsi128_si256(lo), hi, 1), opa);
    store8(dst + x, gray8_opaque(px));
  }
  gray_row_scalar(dst + x, src + (size_t)x * 3, w - x, 3);
}
#endif

// w pixels of `channels` (3: R G B, 4: R G B A, non-premultiplied) interleaved bytes -> w luminance bytes
inline void gray_row(uint8_t* dst, const uint8_t* src, int w, int channels)
{
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) { gray_row_avx2(dst, src, w, channels); return; }
#endif
  gray_row_scalar(dst, src, w, channels);
}

}  // namespace sadhip
