// despair.cpp -- see despair.hpp.  Host logic only; all arithmetic happens in libsadhip.so.
#include "despair.hpp"
#include "../../include/sadhip.h"

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

}  // namespace despair
