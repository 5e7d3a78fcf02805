// despair.cpp -- see despair.hpp.  Host logic only; all arithmetic happens in libsadhip.so.
#include "despair.hpp"
#include "png_io.hpp"
#include "../../include/sadhip.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace despair {

namespace {
std::mutex g_params_mu;                 // params.go:8-11
Parameters g_params{16, 64};            // params.go:13-18

void check(int st, const char* what)
{
  if (st != SADHIP_OK) throw Error(st, std::string(what) + ": " + sadhip_status_string(st) + " (" + sadhip_last_error() + ")");
}

// process-wide context per device, grown on demand
struct Ctx {
  std::mutex mu; sadhip_ctx* h = nullptr; int w = 0, hgt = 0, device = 0;
  ~Ctx() { if (h) sadhip_ctx_destroy(h); }
  sadhip_ctx* get(int dev, int width, int height) {
    std::lock_guard<std::mutex> lk(mu);
    if (!h || dev != device || width > w || height > hgt) {
      if (h) { sadhip_ctx_destroy(h); h = nullptr; }
      const int nw = width > 1920 ? width : 1920, nh = height > 1080 ? height : 1080;
      check(sadhip_ctx_create(dev, nw > w ? nw : w, nh > hgt ? nh : hgt, 1, &h), "sadhip_ctx_create");
      w = nw > w ? nw : w; hgt = nh > hgt ? nh : hgt; device = dev;
    }
    return h;
  }
};
Ctx& ctx() { static Ctx c; return c; }
}  // namespace

void SetDefaultParams(Parameters p) { std::lock_guard<std::mutex> lk(g_params_mu); g_params = p; }
Parameters DefaultParams() { std::lock_guard<std::mutex> lk(g_params_mu); return g_params; }

Pipeline SetupConcurrentSAD(int numWorkers, int device)
{
  if (numWorkers <= 0) numWorkers = (int)std::thread::hardware_concurrency() * 4;      // sad.go:32-34
  Pipeline p{std::make_shared<Chan<InputChunk>>((size_t)numWorkers * 2),               // sad.go:36-37
             std::make_shared<Chan<OutputChunk>>((size_t)numWorkers * 2)};
  std::thread([in = p.in, out = p.out, device] {
    InputChunk c;
    while (in->recv(c)) {                                                              // sad.go:47
      const Parameters prm = DefaultParams();                                          // sad.go:51-53 (per chunk)
      OutputChunk o;
      o.Region = c.Region;
      o.DisparityData.assign((size_t)c.Region.Dx() * c.Region.Dy(), 0);                // sad.go:48-50
      try {
        const Gray& L = *c.Left; const Gray& R = *c.Right;
        check(sadhip_disparity_region_u8(ctx().get(device, L.Rect.Dx(), L.Rect.Dy()), L.Pix.data(), R.Pix.data(),
                                         L.Rect.Dx(), L.Rect.Dy(), L.Stride, R.Stride, prm.BlockSize, prm.MaxDisparity, 0,
                                         c.Region.x0, c.Region.y0, c.Region.x1, c.Region.y1, o.DisparityData.data()),
              "sadhip_disparity_region_u8");
      } catch (const Error&) {
        out->close();                                                                   // surface the failure: consumer sees a short stream
        return;
      }
      out->send(std::move(o));                                                          // sad.go:98-101
    }
    out->close();                                                                       // sad.go:106-110
  }).detach();
  return p;
}

Gray AssembleDisparityMap(Chan<OutputChunk>& out, Rectangle dimensions, int chunks)
{
  Gray m(dimensions);
  OutputChunk c;
  int i = 0;
  while (i < chunks && out.recv(c)) {
    const int w = c.Region.Dx();
    for (int y = 0; y < c.Region.Dy(); ++y)
      for (int x = 0; x < w; ++x)
        m.Pix[(size_t)(c.Region.y0 + y - dimensions.y0) * m.Stride + (c.Region.x0 + x - dimensions.x0)] =
            c.DisparityData[(size_t)y * w + x];                                         // sad.go:186-197
    ++i;                                                                                // every chunk is written (no sad.go:181-184 hole)
  }
  return m;
}

Gray RunSad(const Gray& left, const Gray& right, int blockSize, int maxDisparity, int device)
{
  SetDefaultParams(Parameters{blockSize, maxDisparity});                                // sad.go:123-126
  Gray out(left.Rect);
  check(sadhip_disparity_u8(ctx().get(device, left.Rect.Dx(), left.Rect.Dy()), left.Pix.data(), right.Pix.data(),
                            left.Rect.Dx(), left.Rect.Dy(), left.Stride, right.Stride, blockSize, maxDisparity, 0,
                            out.Pix.data(), out.Stride),
        "sadhip_disparity_u8");
  return out;
}

int SumAbsoluteDifferences(const Gray& left, const Gray& right, int leftX, int leftY, int rightX, int rightY, int blockSize)
{
  const int32_t q[4] = {leftX, leftY, rightX, rightY};
  int32_t r = 0;
  check(sadhip_sad_windows_u8(ctx().get(0, left.Rect.Dx(), left.Rect.Dy()), left.Pix.data(), left.Rect.Dx(), left.Rect.Dy(), left.Stride,
                              right.Pix.data(), right.Rect.Dx(), right.Rect.Dy(), right.Stride, blockSize, 1, q, &r),
        "sadhip_sad_windows_u8");
  return r;
}

// ---- png.go ------------------------------------------------------------------------------------
Gray LoadPNG(const std::string& filename, LoadPNGOptions opt)
{
  PngImage im;
  std::string err;
  if (!read_png(filename, im, &err)) throw Error(SADHIP_ERR_ARG, "LoadPNG " + filename + ": " + err);
  Gray g(im.w, im.h);
  const size_t npx = (size_t)im.w * im.h;
  switch (im.kind) {
    case GoImage::Gray:                                         // png.go:31-32 -> gray.go:15-17
      g.Pix = im.samples;
      break;
    case GoImage::RGBA:                                         // png.go:33-34 -> gray.go:20-40
      if (opt.refbug_rgba) { std::fill(g.Pix.begin(), g.Pix.end(), 0); break; }   // 8-bit channels >> 24: always 0
      [[fallthrough]];
    case GoImage::NRGBA:
    case GoImage::Paletted:                                     // png.go:35-36 -> gray.go:43-58, on the device
      check(sadhip_gray_u8(ctx().get(opt.device, im.w, im.h), im.samples.data(), im.channels, im.w, im.h,
                           im.w * im.channels, g.Pix.data(), g.Stride), "sadhip_gray_u8");
      break;
    case GoImage::Gray16:                                       // color.Gray16.RGBA() = (y, y, y): formula keeps the high byte
      for (size_t i = 0; i < npx; ++i) {
        const uint32_t y = (uint32_t)im.samples[2 * i] << 8 | im.samples[2 * i + 1];
        g.Pix[i] = (uint8_t)((19595u * y + 38470u * y + 7471u * y + (1u << 15)) >> 24);
      }
      break;
    case GoImage::RGBA64:
    case GoImage::NRGBA64:                                      // 16-bit channels, NRGBA64 premultiplies r*a/0xffff
      for (size_t i = 0; i < npx; ++i) {
        const uint8_t* s = im.samples.data() + i * 2 * im.channels;
        uint32_t r = (uint32_t)s[0] << 8 | s[1], gg = (uint32_t)s[2] << 8 | s[3], b = (uint32_t)s[4] << 8 | s[5];
        if (im.channels == 4) {
          const uint32_t a = (uint32_t)s[6] << 8 | s[7];
          r = r * a / 0xffff; gg = gg * a / 0xffff; b = b * a / 0xffff;
        }
        g.Pix[i] = (uint8_t)((19595u * r + 38470u * gg + 7471u * b + (1u << 15)) >> 24);
      }
      break;
  }
  return g;
}

void SavePNG(const std::string& filename, const Gray& img)
{
  std::string err;
  if (!write_png_gray8(filename, img.Pix.data(), img.Rect.Dx(), img.Rect.Dy(), img.Stride, &err))
    throw Error(SADHIP_ERR_ARG, "SavePNG " + filename + ": " + err);
}

}  // namespace despair
