// png_io.hpp -- PNG decode/encode for the host mirror of pkg/despair/png.go.
// CPU-side work either side of the hot path: zlib is the only dependency.  Non-interlaced PNGs of
// every colour type and bit depth 8 (plus 1/2/4-bit gray and palette, 16-bit truncated the way
// Go's colour model does); Adam7 interlacing is rejected with an error.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace despair {

// Which Go image type png.Decode would hand to LoadPNG's type switch (png.go:30-37).
enum class GoImage {
  Gray,      // colour type 0, depth <= 8, no tRNS                 -> *image.Gray
  Gray16,    // colour type 0, depth 16                            -> *image.Gray16
  RGBA,      // colour type 2, depth 8, no tRNS (opaque)           -> *image.RGBA
  NRGBA,     // colour type 6 depth 8; type 4 depth 8; types 0/2 depth 8 with tRNS -> *image.NRGBA
  RGBA64,    // colour type 2, depth 16                            -> *image.RGBA64
  NRGBA64,   // colour types 4/6 depth 16                          -> *image.NRGBA64
  Paletted,  // colour type 3                                      -> *image.Paletted
};

// Decoded picture, normalised for the caller: `samples` holds `channels` interleaved samples per
// pixel (1 = gray, 3 = RGB, 4 = non-premultiplied RGBA), 8 bits each unless depth16 (then two
// bytes per sample, big endian as in the file).  Palette and low-bit-depth pictures are expanded.
struct PngImage {
  int w = 0, h = 0, channels = 0;
  bool depth16 = false;
  GoImage kind = GoImage::Gray;
  std::vector<uint8_t> samples;
};

// Both return false and fill *err on failure; they never throw.
bool read_png(const std::string& path, PngImage& out, std::string* err);
bool decode_png(const uint8_t* data, size_t n, PngImage& out, std::string* err);
// 8-bit gray, filter 0, zlib level 1 (png.go:56-58: png.BestSpeed).
bool write_png_gray8(const std::string& path, const uint8_t* pix, int w, int h, int stride, std::string* err);
bool encode_png_gray8(const uint8_t* pix, int w, int h, int stride, std::vector<uint8_t>& out, std::string* err);

}  // namespace despair
