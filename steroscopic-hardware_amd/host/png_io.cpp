// png_io.cpp -- see png_io.hpp.  Written against the PNG specification (chunk layout, the five
// scanline filters, zlib streams); image typing follows what Go's image/png hands to
// pkg/despair/png.go so that LoadPNG can take the same branch the reference would.
#include "png_io.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace despair {
namespace {

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

bool fail(std::string* err, const std::string& what)
{
  if (err) *err = what;
  return false;
}

int paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

void append_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n)
{
  put32(out, (uint32_t)n);
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  if (n) out.insert(out.end(), data, data + n);
  put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

}  // namespace

bool decode_png(const uint8_t* data, size_t n, PngImage& out, std::string* err)
{
  if (n < 8 || memcmp(data, kSig, 8) != 0) return fail(err, "not a PNG file");
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  bool seen_ihdr = false, seen_iend = false, has_trns = false;
  while (pos + 12 <= n && !seen_iend) {
    const uint32_t len = be32(data + pos);
    if (len > n - pos - 12) return fail(err, "truncated chunk");
    const uint8_t* type = data + pos + 4;
    const uint8_t* body = data + pos + 8;
    if (crc32(0L, type, len + 4) != be32(body + len)) return fail(err, "chunk CRC mismatch");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail(err, "bad IHDR");
      w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      if (body[10] != 0 || body[11] != 0) return fail(err, "unknown compression/filter method");
      seen_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!memcmp(type, "tRNS", 4)) {
      trns.assign(body, body + len); has_trns = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      seen_iend = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!seen_ihdr || !seen_iend) return fail(err, "missing IHDR/IEND");
  if (w <= 0 || h <= 0 || (long long)w * h > (1LL << 30)) return fail(err, "bad dimensions");
  if (interlace != 0) return fail(err, "Adam7 interlaced PNG is not supported");
  int nch;
  switch (ctype) {
    case 0: nch = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return fail(err, "bad bit depth"); break;
    case 2: nch = 3; if (depth != 8 && depth != 16) return fail(err, "bad bit depth"); break;
    case 3: nch = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8) return fail(err, "bad bit depth"); break;
    case 4: nch = 2; if (depth != 8 && depth != 16) return fail(err, "bad bit depth"); break;
    case 6: nch = 4; if (depth != 8 && depth != 16) return fail(err, "bad bit depth"); break;
    default: return fail(err, "bad colour type");
  }
  if (ctype == 3 && (plte.empty() || plte.size() % 3 != 0)) return fail(err, "palette image without PLTE");
  if (has_trns && depth == 16) return fail(err, "16-bit PNG with tRNS is not supported");

  // ---- inflate + unfilter ---------------------------------------------------------------------
  const size_t bpp_bits = (size_t)nch * depth;
  const size_t row_bytes = ((size_t)w * bpp_bits + 7) / 8;
  const size_t fbpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;          // filter byte distance
  // deflate expands at most ~1032:1: a header that promises more than the IDAT bytes can hold is
  // rejected before anything of that size is allocated
  if ((row_bytes + 1) * (size_t)h > idat.size() * 1100 + 65536) return fail(err, "zlib stream does not match the image size");
  std::vector<uint8_t> raw((row_bytes + 1) * (size_t)h);
  uLongf raw_len = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || raw_len != raw.size()) return fail(err, "zlib stream does not match the image size");
  std::vector<uint8_t> img(row_bytes * (size_t)h), zero(row_bytes, 0);
  for (int y = 0; y < h; ++y) {
    const uint8_t* src = raw.data() + (row_bytes + 1) * (size_t)y;
    uint8_t* cur = img.data() + row_bytes * (size_t)y;
    const uint8_t* up = y ? cur - row_bytes : zero.data();
    const int ft = src[0];
    ++src;
    for (size_t i = 0; i < row_bytes; ++i) {
      const int a = i >= fbpp ? cur[i - fbpp] : 0, b = up[i], c = i >= fbpp ? up[i - fbpp] : 0;
      int v = src[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return fail(err, "bad scanline filter");
      }
      cur[i] = (uint8_t)v;
    }
  }

  // ---- normalise: what Go's decoder would return, as interleaved samples -------------------------
  out = PngImage{};
  out.w = w; out.h = h;
  auto sample = [&](const uint8_t* row, int idx) -> int {      // sample idx of a row, depth < 8 unpacked
    if (depth == 8) return row[idx];
    const int per = 8 / depth, byte = idx / per, shift = 8 - depth * (idx % per + 1);
    return (row[byte] >> shift) & ((1 << depth) - 1);
  };
  if (depth == 16) {
    out.depth16 = true;
    if (ctype == 0) { out.channels = 1; out.kind = GoImage::Gray16; out.samples = img; }
    else if (ctype == 2) { out.channels = 3; out.kind = GoImage::RGBA64; out.samples = img; }
    else if (ctype == 6) { out.channels = 4; out.kind = GoImage::NRGBA64; out.samples = img; }
    else {                                                       // gray + alpha -> (y, y, y, a)
      out.channels = 4; out.kind = GoImage::NRGBA64; out.samples.resize((size_t)w * h * 8);
      for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t* s = img.data() + i * 4; uint8_t* d = out.samples.data() + i * 8;
        d[0] = d[2] = d[4] = s[0]; d[1] = d[3] = d[5] = s[1]; d[6] = s[2]; d[7] = s[3];
      }
    }
    return true;
  }
  if (ctype == 0 && !has_trns) {                                 // *image.Gray, low depths scaled to 8 bits
    out.channels = 1; out.kind = GoImage::Gray; out.samples.resize((size_t)w * h);
    const int mul = depth == 8 ? 1 : 255 / ((1 << depth) - 1);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) out.samples[(size_t)y * w + x] = (uint8_t)(sample(img.data() + row_bytes * y, x) * mul);
    return true;
  }
  if (ctype == 2 && !has_trns) { out.channels = 3; out.kind = GoImage::RGBA; out.samples = img; return true; }
  out.channels = 4; out.samples.resize((size_t)w * h * 4);
  out.kind = ctype == 3 ? GoImage::Paletted : GoImage::NRGBA;
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = img.data() + row_bytes * y;
    for (int x = 0; x < w; ++x) {
      uint8_t* d = out.samples.data() + ((size_t)y * w + x) * 4;
      if (ctype == 6) { memcpy(d, row + 4 * x, 4); }
      else if (ctype == 4) { d[0] = d[1] = d[2] = row[2 * x]; d[3] = row[2 * x + 1]; }
      else if (ctype == 3) {
        const size_t i = (size_t)sample(row, x);
        if (i * 3 + 2 >= plte.size()) return fail(err, "palette index out of range");
        d[0] = plte[i * 3]; d[1] = plte[i * 3 + 1]; d[2] = plte[i * 3 + 2];
        d[3] = i < trns.size() ? trns[i] : 255;
      } else if (ctype == 2) {                                   // RGB + tRNS colour key (16-bit fields, low byte used)
        memcpy(d, row + 3 * x, 3);
        const bool key = trns.size() >= 6 && d[0] == trns[1] && d[1] == trns[3] && d[2] == trns[5];
        d[3] = key ? 0 : 255;
      } else {                                                   // gray + tRNS key
        const int v = sample(row, x), mul = depth == 8 ? 1 : 255 / ((1 << depth) - 1);
        d[0] = d[1] = d[2] = (uint8_t)(v * mul);
        const bool key = trns.size() >= 2 && v == ((trns[0] << 8 | trns[1]) & ((1 << depth) - 1));
        d[3] = key ? 0 : 255;
      }
    }
  }
  return true;
}

bool read_png(const std::string& path, PngImage& out, std::string* err)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(err, "open " + path + ": cannot open file");
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t got;
  while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
  fclose(f);
  return decode_png(buf.data(), buf.size(), out, err);
}

bool encode_png_gray8(const uint8_t* pix, int w, int h, int stride, std::vector<uint8_t>& out, std::string* err)
{
  if (!pix || w <= 0 || h <= 0 || stride < w) return fail(err, "bad image");
  out.assign(kSig, kSig + 8);
  uint8_t ihdr[13] = {0};
  ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
  ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
  ihdr[8] = 8; ihdr[9] = 0;                                      // 8-bit gray, deflate, adaptive filter set, no interlace
  append_chunk(out, "IHDR", ihdr, 13);
  std::vector<uint8_t> raw(((size_t)w + 1) * h);
  for (int y = 0; y < h; ++y) {
    raw[((size_t)w + 1) * y] = 0;                                // filter type None
    memcpy(raw.data() + ((size_t)w + 1) * y + 1, pix + (size_t)stride * y, (size_t)w);
  }
  uLongf zl = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zl);
  if (compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) != Z_OK) return fail(err, "zlib compress failed");
  append_chunk(out, "IDAT", z.data(), zl);
  append_chunk(out, "IEND", nullptr, 0);
  return true;
}

bool write_png_gray8(const std::string& path, const uint8_t* pix, int w, int h, int stride, std::string* err)
{
  std::vector<uint8_t> buf;
  if (!encode_png_gray8(pix, w, h, stride, buf, err)) return false;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return fail(err, "open " + path + ": cannot create file");
  const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  return (fclose(f) == 0 && ok) ? true : fail(err, "write " + path + " failed");
}

}  // namespace despair
