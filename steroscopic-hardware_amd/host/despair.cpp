// despair.cpp -- see despair.hpp.  Host logic only; all arithmetic happens in libsadhip.so.
#include "despair.hpp"
#include "png_io.hpp"
#include "../../include/sadhip.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <shared_mutex>
#include <thread>

namespace despair {

namespace {
std::mutex g_params_mu;                 // params.go:8-11
Parameters g_params{16, 64};            // params.go:13-18

void check(int st, const char* what)
{
  if (st != SADHIP_OK) throw Error(st, std::string(what) + ": " + sadhip_status_string(st) + " (" + sadhip_last_error() + ")");
}

// Process-wide contexts, one per device, grown on demand.  A caller HOLDS A SHARED LOCK on its
// device's entry for the whole sadhip_* call (Lease); growing takes the exclusive lock, so a context
// is never destroyed while any thread is inside the library with its handle -- OutputCamera and
// RunSad callers are concurrent in the reference (pkg/camera/output.go:37, examples/run.go:30).
// A pipeline on another device has its own entry: nothing is torn down on a device change.
struct DeviceCtx {
  std::shared_mutex mu;
  sadhip_ctx* h = nullptr;
  int w = 0, hgt = 0;
  ~DeviceCtx() { if (h) sadhip_ctx_destroy(h); }
};

struct Registry {
  std::mutex mu;
  std::map<int, std::unique_ptr<DeviceCtx>> by_device;
  DeviceCtx& entry(int dev) {
    std::lock_guard<std::mutex> lk(mu);
    auto& e = by_device[dev];
    if (!e) e.reset(new DeviceCtx());
    return *e;
  }
};
Registry& registry()
{
  // the header this file was compiled against and the library the process loaded must be the same ABI
  static const bool abi_ok = [] {
    if (sadhip_abi_version() != SADHIP_ABI_VERSION)
      throw std::runtime_error("despair: libsadhip ABI " + std::to_string(sadhip_abi_version()) + " loaded, built against " +
                               std::to_string(SADHIP_ABI_VERSION));
    return true;
  }();
  (void)abi_ok;
  static Registry r;
  return r;
}

class Lease {
 public:
  Lease(int dev, int width, int height) : e_(registry().entry(dev)), lk_(e_.mu, std::defer_lock)
  {
    for (;;) {
      lk_.lock();
      if (e_.h && width <= e_.w && height <= e_.hgt) return;           // shared lock stays held until ~Lease
      lk_.unlock();
      std::unique_lock<std::shared_mutex> grow(e_.mu);                 // waits for every in-flight call
      if (e_.h && width <= e_.w && height <= e_.hgt) continue;         // someone else grew it meanwhile
      if (e_.h) { sadhip_ctx_destroy(e_.h); e_.h = nullptr; }
      const int nw = std::max({width, e_.w, 1920}), nh = std::max({height, e_.hgt, 1080});
      check(sadhip_ctx_create(dev, nw, nh, 1, &e_.h), "sadhip_ctx_create");
      e_.w = nw; e_.hgt = nh;
    }
  }
  sadhip_ctx* get() const { return e_.h; }
 private:
  DeviceCtx& e_;
  std::shared_lock<std::shared_mutex> lk_;
};

thread_local float t_kernel_ms = 0.f, t_total_ms = 0.f;
void note_timing(sadhip_ctx* h) { (void)sadhip_last_timing(h, &t_kernel_ms, &t_total_ms); }

int dev_or_default(int device) { return device >= 0 ? device : DefaultDevice(); }
}  // namespace

int DefaultDevice()
{
  static const int d = [] { const char* e = getenv("SADHIP_DEVICE"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : v; }();
  return d;
}

void SetDefaultParams(Parameters p) { std::lock_guard<std::mutex> lk(g_params_mu); g_params = p; }
Parameters DefaultParams() { std::lock_guard<std::mutex> lk(g_params_mu); return g_params; }
void LastTiming(float* kernel_ms, float* total_ms) { if (kernel_ms) *kernel_ms = t_kernel_ms; if (total_ms) *total_ms = t_total_ms; }

Pipeline SetupConcurrentSAD(int numWorkers, int device)
{
  if (numWorkers <= 0) numWorkers = (int)std::thread::hardware_concurrency() * 4;      // sad.go:32-34
  device = dev_or_default(device);
  // sad.go:36-37 gives both channels 2 x numWorkers slots and every one of its numWorkers goroutines holds a chunk while it
  // waits to send: 5 x numWorkers chunks can be under way before anyone reads the output -- and the reference's callers rely on
  // it: examples/run.stream.go:46-58 feeds all ceil(Dy / max(1, Dy / (4 numWorkers))) chunks of a frame (160 for its 640x480
  // picture with 32 workers) BEFORE it starts to read.  One device worker holds no such crowd, so the output channel gets
  // the goroutines' share: 3 x numWorkers.
  Pipeline p{std::make_shared<Chan<InputChunk>>((size_t)numWorkers * 2),
             std::make_shared<Chan<OutputChunk>>((size_t)numWorkers * 3)};
  std::thread([in = p.in, out = p.out, device] {
    constexpr size_t kMaxGroup = 512;                                                  // chunks answered by one device call
    InputChunk c, next;
    bool failed = false, have_next = false;
    for (;;) {
      if (have_next) { c = std::move(next); have_next = false; }
      else if (!in->recv(c)) break;                                                    // sad.go:47
      if (failed) continue;                                                            // drain: a feeder blocked in send() must get out
      // the chunks already queued behind this one that continue it downwards
      std::vector<InputChunk> group;
      group.push_back(std::move(c));
      while (group.size() < kMaxGroup && in->try_recv(next)) {
        const InputChunk& f = group.front(); const InputChunk& l = group.back();
        if (next.Left == f.Left && next.Right == f.Right && next.Region.x0 == f.Region.x0 && next.Region.x1 == f.Region.x1 &&
            next.Region.y0 == l.Region.y1 && next.Region.Dy() > 0 && l.Region.Dy() > 0) {
          group.push_back(std::move(next));
        } else { have_next = true; break; }                                           // starts the next group
      }
      const Parameters prm = DefaultParams();                                          // sad.go:51-53 (per group: see despair.hpp)
      const Rectangle all{group.front().Region.x0, group.front().Region.y0, group.front().Region.x1, group.back().Region.y1};
      std::vector<uint8_t> data;
      try {
        data.assign((size_t)all.Dx() * all.Dy(), 0);                                   // sad.go:48-50
        if (!data.empty()) {
          const Gray& L = *group.front().Left; const Gray& R = *group.front().Right;
          Lease ctx(device, L.Rect.Dx(), L.Rect.Dy());
          // each image with its OWN size: the two pictures are decoded independently (pkg/camera/output.go:134-162) and
          // the reference clamps / breaks per image (sad.go:212-218,225-227,231-233) -- never w x h of a smaller R
          check(sadhip_disparity_lr_u8(ctx.get(), L.Pix.data(), L.Rect.Dx(), L.Rect.Dy(), L.Stride,
                                       R.Pix.data(), R.Rect.Dx(), R.Rect.Dy(), R.Stride, prm.BlockSize, prm.MaxDisparity, 0,
                                       all.x0, all.y0, all.x1, all.y1, data.data(), all.Dx()),
                "sadhip_disparity_lr_u8");
        }
      } catch (...) {
        out->fail(std::current_exception());                                           // the consumer gets the error, not a short stream
        failed = true;
        continue;
      }
      for (const InputChunk& g : group) {                                              // sad.go:98-101, one answer per chunk
        OutputChunk o;
        o.Region = g.Region;
        const size_t row = (size_t)all.Dx();
        o.DisparityData.assign(data.begin() + (size_t)(g.Region.y0 - all.y0) * row, data.begin() + (size_t)(g.Region.y1 - all.y0) * row);
        try { out->send(std::move(o)); }
        catch (const std::logic_error&) { failed = true; break; }                      // consumer closed the output: drain and leave
      }
    }
    if (!failed) out->close();                                                         // sad.go:106-110
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
  if (i < chunks) {                                                                     // never a silently short map
    if (std::exception_ptr e = out.error()) std::rethrow_exception(e);
    throw Error(SADHIP_ERR_ARG, "AssembleDisparityMap: output channel closed after " + std::to_string(i) + " of " +
                                std::to_string(chunks) + " chunks");
  }
  return m;
}

Gray RunSad(const Gray& left, const Gray& right, int blockSize, int maxDisparity, int device)
{
  SetDefaultParams(Parameters{blockSize, maxDisparity});                                // sad.go:123-126
  return RunSadFrame(left, right, Parameters{blockSize, maxDisparity}, device);
}

Gray RunSadFrame(const Gray& left, const Gray& right, Parameters p, int device)
{
  const int blockSize = p.BlockSize, maxDisparity = p.MaxDisparity;
  Gray out(left.Rect);                                                                  // sad.go:135,168: the LEFT image's rectangle
  Lease ctx(dev_or_default(device), left.Rect.Dx(), left.Rect.Dy());
  if (right.Rect.Dx() == left.Rect.Dx() && right.Rect.Dy() == left.Rect.Dy())
    check(sadhip_disparity_u8(ctx.get(), left.Pix.data(), right.Pix.data(),
                              left.Rect.Dx(), left.Rect.Dy(), left.Stride, right.Stride, blockSize, maxDisparity, 0,
                              out.Pix.data(), out.Stride),
          "sadhip_disparity_u8");
  else                                                                                  // sizes differ: the reference's clamps (sad.go:212-233)
    check(sadhip_disparity_lr_u8(ctx.get(), left.Pix.data(), left.Rect.Dx(), left.Rect.Dy(), left.Stride,
                                 right.Pix.data(), right.Rect.Dx(), right.Rect.Dy(), right.Stride, blockSize, maxDisparity, 0,
                                 0, 0, left.Rect.Dx(), left.Rect.Dy(), out.Pix.data(), out.Stride),
          "sadhip_disparity_lr_u8");
  note_timing(ctx.get());
  return out;
}

int SumAbsoluteDifferences(const Gray& left, const Gray& right, int leftX, int leftY, int rightX, int rightY, int blockSize)
{
  const int32_t q[4] = {leftX, leftY, rightX, rightY};
  int32_t r = 0;
  Lease ctx(DefaultDevice(), std::max(left.Rect.Dx(), right.Rect.Dx()), std::max(left.Rect.Dy(), right.Rect.Dy()));
  check(sadhip_sad_windows_u8(ctx.get(), left.Pix.data(), left.Rect.Dx(), left.Rect.Dy(), left.Stride,
                              right.Pix.data(), right.Rect.Dx(), right.Rect.Dy(), right.Stride, blockSize, 1, q, &r),
        "sadhip_sad_windows_u8");
  return r;
}

// ---- png.go ------------------------------------------------------------------------------------
Picture LoadPicture(const std::string& filename)
{
  PngImage im;
  std::string err;
  if (!read_png(filename, im, &err)) throw Error(SADHIP_ERR_ARG, "LoadPNG " + filename + ": " + err);
  Picture p;
  p.w = im.w; p.h = im.h; p.channels = im.channels; p.kind = (int)im.kind; p.depth16 = im.depth16;
  p.samples = std::move(im.samples);
  return p;
}

Gray ToGray(const Picture& im, LoadPNGOptions opt)
{
  Gray g(im.w, im.h);
  const size_t npx = (size_t)im.w * im.h;
  switch ((GoImage)im.kind) {
    case GoImage::Gray:                                         // png.go:31-32 -> gray.go:15-17
      g.Pix = im.samples;
      break;
    case GoImage::RGBA:                                         // png.go:33-34 -> gray.go:20-40
      if (opt.refbug_rgba) { std::fill(g.Pix.begin(), g.Pix.end(), 0); break; }   // 8-bit channels >> 24: always 0
      [[fallthrough]];
    case GoImage::NRGBA:
    case GoImage::Paletted: {                                   // png.go:35-36 -> gray.go:43-58, on the device
      Lease ctx(dev_or_default(opt.device), im.w, im.h);
      check(sadhip_gray_u8(ctx.get(), im.samples.data(), im.channels, im.w, im.h,
                           im.w * im.channels, g.Pix.data(), g.Stride), "sadhip_gray_u8");
      break;
    }
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

Gray LoadPNG(const std::string& filename, LoadPNGOptions opt) { return ToGray(LoadPicture(filename), opt); }

Gray RunSad(const Picture& left, const Picture& right, int blockSize, int maxDisparity, int device, int* uploads_out)
{
  SetDefaultParams(Parameters{blockSize, maxDisparity});                                // sad.go:123-126
  return RunSadImages(left, right, Parameters{blockSize, maxDisparity}, device, uploads_out);
}

static std::atomic<unsigned> g_host_gray{0};
void SetHostGray(bool on) { g_host_gray.store(on ? SADHIP_FLAG_HOST_GRAY : 0u); }

Gray RunSadImages(const Picture& left, const Picture& right, Parameters p, int device, int* uploads_out)
{
  const int blockSize = p.BlockSize, maxDisparity = p.MaxDisparity;
  // the Go shim's RunSadImages (host/go/sad_hip.go): two 8-bit RGB(A) pictures of one size and channel count are ONE
  // device call; gray pictures go as they are; anything else -- other pixel formats, pictures of different sizes or
  // channel counts (left.png / right.png are uploaded independently) -- is converted per picture (ToGray = GrayOf) and
  // sent through the lr entry, each picture with its own size
  if (left.w == right.w && left.h == right.h && left.device_gray_ok() && right.device_gray_ok() && left.channels == right.channels) {
    Gray out(left.w, left.h);
    Lease ctx(dev_or_default(device), left.w, left.h);
    check(sadhip_disparity_rgba_u8(ctx.get(), left.samples.data(), right.samples.data(), left.channels, left.w, left.h,
                                   left.w * left.channels, right.w * right.channels, blockSize, maxDisparity, g_host_gray.load(),
                                   out.Pix.data(), out.Stride),
          "sadhip_disparity_rgba_u8");
    note_timing(ctx.get());
    if (uploads_out) *uploads_out = 2;                                                  // one per image; gray never visits the host
    return out;
  }
  LoadPNGOptions opt; opt.device = device;
  int up = 2;                                                                           // the two gray planes of RunSad
  if (left.device_gray_ok() || (GoImage)left.kind == GoImage::Paletted) ++up;
  if (right.device_gray_ok() || (GoImage)right.kind == GoImage::Paletted) ++up;
  if (uploads_out) *uploads_out = up;
  return RunSadFrame(ToGray(left, opt), ToGray(right, opt), p, device);
}

void SavePNG(const std::string& filename, const Gray& img)
{
  std::string err;
  if (!write_png_gray8(filename, img.Pix.data(), img.Rect.Dx(), img.Rect.Dy(), img.Stride, &err))
    throw Error(SADHIP_ERR_ARG, "SavePNG " + filename + ": " + err);
}

// ---- page-locked frames --------------------------------------------------------------------------
HostFrames::HostFrames(int n, int w, int h) : n_(n), w_(w), h_(h)
{
  if (n < 1 || w < 1 || h < 1) throw Error(SADHIP_ERR_DIMS, "HostFrames: n, w, h must be positive");
  void* p = nullptr;
  check(sadhip_host_alloc((size_t)n * w * h, &p), "sadhip_host_alloc");
  base_ = static_cast<uint8_t*>(p);
}
HostFrames::~HostFrames() { if (base_) sadhip_host_free(base_); }

// ---- several devices -----------------------------------------------------------------------------
DevicePool::DevicePool(const std::vector<int>& devices, int max_w, int max_h)
{
  check(sadhip_pool_create(devices.data(), (int)devices.size(), max_w, max_h, &p_), "sadhip_pool_create");
}
DevicePool::~DevicePool() { if (p_) sadhip_pool_destroy(p_); }
int DevicePool::size() const { int n = 0; check(sadhip_pool_size(p_, &n), "sadhip_pool_size"); return n; }

std::vector<Gray> DevicePool::RunSadBatch(const std::vector<std::shared_ptr<const Gray>>& lefts,
                                          const std::vector<std::shared_ptr<const Gray>>& rights, int blockSize, int maxDisparity)
{
  if (lefts.size() != rights.size()) throw Error(SADHIP_ERR_BATCH, "RunSadBatch: lefts/rights differ in length");
  SetDefaultParams(Parameters{blockSize, maxDisparity});
  std::vector<Gray> outs;
  if (lefts.empty()) return outs;
  const Rectangle r = lefts[0]->Rect;
  const int ls = lefts[0]->Stride, rs = rights[0]->Stride;
  std::vector<const uint8_t*> pl, pr;
  std::vector<uint8_t*> po;
  outs.reserve(lefts.size());
  for (size_t i = 0; i < lefts.size(); ++i) {
    if (lefts[i]->Rect.Dx() != r.Dx() || lefts[i]->Rect.Dy() != r.Dy() || rights[i]->Rect.Dx() != r.Dx() || rights[i]->Rect.Dy() != r.Dy() ||
        lefts[i]->Stride != ls || rights[i]->Stride != rs)
      throw Error(SADHIP_ERR_DIMS, "RunSadBatch: frames of one batch must share size and stride");
    outs.emplace_back(r);
    pl.push_back(lefts[i]->Pix.data()); pr.push_back(rights[i]->Pix.data());
  }
  for (Gray& g : outs) po.push_back(g.Pix.data());
  check(sadhip_pool_batch_u8(p_, (int)lefts.size(), pl.data(), pr.data(), r.Dx(), r.Dy(), ls, rs, blockSize, maxDisparity, 0,
                             po.data(), r.Dx()), "sadhip_pool_batch_u8");
  return outs;
}

void DevicePool::RunSadBatch(const HostFrames& lefts, const HostFrames& rights, HostFrames& outs, int blockSize, int maxDisparity)
{
  const int n = lefts.size(), w = lefts.width(), h = lefts.height();
  if (rights.size() != n || outs.size() != n || rights.width() != w || outs.width() != w || rights.height() != h || outs.height() != h)
    throw Error(SADHIP_ERR_DIMS, "RunSadBatch: lefts / rights / outs must share length and frame size");
  SetDefaultParams(Parameters{blockSize, maxDisparity});
  std::vector<const uint8_t*> pl((size_t)n), pr((size_t)n);
  std::vector<uint8_t*> po((size_t)n);
  for (int i = 0; i < n; ++i) { pl[(size_t)i] = lefts.frame(i); pr[(size_t)i] = rights.frame(i); po[(size_t)i] = outs.frame(i); }
  check(sadhip_pool_batch_u8(p_, n, pl.data(), pr.data(), w, h, w, w, blockSize, maxDisparity, 0, po.data(), w), "sadhip_pool_batch_u8");
}

}  // namespace despair
