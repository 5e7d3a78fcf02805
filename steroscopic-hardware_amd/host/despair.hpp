// despair.hpp -- C++ mirror of the Go API of pkg/despair (the reference's compiled host side),
// implemented over the C ABI of libsadhip.so (include/sadhip.h).  Same names, argument meaning
// and ownership rules as the reference so callers (pkg/camera/output.go, examples/run*.go) map
// one to one; the cgo shim in go/ is the same logic in Go.
//
//   Go (pkg/despair)                                   here
//   type Parameters{BlockSize,MaxDisparity}            despair::Parameters            params.go:32-37
//   SetDefaultParams / DefaultParams                   despair::SetDefaultParams/...  params.go:21-30
//   type InputChunk / OutputChunk                      despair::InputChunk/OutputChunk sad.go:11-21
//   SetupConcurrentSAD(numWorkers)                     despair::SetupConcurrentSAD    sad.go:29-113
//   AssembleDisparityMap(out, dims, chunks)            despair::AssembleDisparityMap  sad.go:172-202
//   RunSad(left, right, blockSize, maxDisparity)       despair::RunSad                sad.go:119-169
//   SumAbsoluteDifferences(...)                        despair::SumAbsoluteDifferences sad.go:205-244
//   LoadPNG / MustLoadPNG / SavePNG / MustSavePNG      despair::LoadPNG/...           png.go:10-76
//
// Beyond the reference's surface:
//   LoadPicture + RunSad(Picture, Picture)             decoded PNG samples stay in memory; an RGB(A) pair
//                                                      is ONE sadhip_disparity_rgba_u8 call (one upload per
//                                                      image, gray on the device, no host gray plane)
//   DevicePool                                         sadhip_pool_*: frames round-robin over several devices
//
// Deliberate differences: AssembleDisparityMap writes every chunk (the
// reference drops the last one, sad.go:181-184); errors are exceptions carrying the C-ABI status
// instead of panics -- a device error inside the pipeline worker reaches the consumer
// (AssembleDisparityMap throws it; never a silently short, partly black map); there is no CPU
// fallback; LoadPNG of an opaque 8-bit RGB file returns its luminance (the reference returns all
// zeros there, gray.go:20-40 -- LoadPNGOptions::refbug_rgba reproduces that).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

struct sadhip_pool;

namespace despair {

struct Rectangle { int x0 = 0, y0 = 0, x1 = 0, y1 = 0; int Dx() const { return x1 - x0; } int Dy() const { return y1 - y0; } };

// image.Gray with Rect.Min == (0,0): Pix[y*Stride + x]
struct Gray {
  std::vector<uint8_t> Pix; int Stride = 0; Rectangle Rect;
  Gray() = default;
  explicit Gray(Rectangle r) : Pix((size_t)r.Dx() * r.Dy()), Stride(r.Dx()), Rect(r) {}
  Gray(int w, int h) : Gray(Rectangle{0, 0, w, h}) {}
};

struct Parameters { int BlockSize = 16; int MaxDisparity = 64; };
void SetDefaultParams(Parameters p);
Parameters DefaultParams();

struct InputChunk { std::shared_ptr<const Gray> Left, Right; Rectangle Region; };
struct OutputChunk { std::vector<uint8_t> DisparityData; Rectangle Region; };

class Error : public std::runtime_error {
 public:
  Error(int status, const std::string& what) : std::runtime_error(what), status(status) {}
  int status;
};

// Buffered channel with Go semantics: send blocks when full, recv returns false once closed and
// drained.  fail() is the pipeline's error slot: it closes the channel and keeps the exception for
// the consumer (error()).
template <typename T>
class Chan {
 public:
  explicit Chan(size_t cap) : cap_(cap ? cap : 1) {}
  void send(T v) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_send_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
    if (closed_) throw std::logic_error("send on closed channel");
    q_.push_back(std::move(v));
    cv_recv_.notify_one();
  }
  bool recv(T& out) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_recv_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    out = std::move(q_.front());
    q_.pop_front();
    cv_send_.notify_one();
    return true;
  }
  // what is queued right now, without waiting (the device worker gathers the chunks of a frame that are already there)
  bool try_recv(T& out) {
    std::lock_guard<std::mutex> lk(mu_);
    if (q_.empty()) return false;
    out = std::move(q_.front());
    q_.pop_front();
    cv_send_.notify_one();
    return true;
  }
  void close() { std::lock_guard<std::mutex> lk(mu_); closed_ = true; cv_recv_.notify_all(); cv_send_.notify_all(); }
  void fail(std::exception_ptr e) { std::lock_guard<std::mutex> lk(mu_); err_ = e; closed_ = true; cv_recv_.notify_all(); cv_send_.notify_all(); }
  std::exception_ptr error() { std::lock_guard<std::mutex> lk(mu_); return err_; }
 private:
  std::mutex mu_; std::condition_variable cv_send_, cv_recv_; std::deque<T> q_; size_t cap_; bool closed_ = false;
  std::exception_ptr err_;
};

struct Pipeline { std::shared_ptr<Chan<InputChunk>> in; std::shared_ptr<Chan<OutputChunk>> out; };

// Device used when a call does not name one: the environment variable SADHIP_DEVICE, read once
// (default 0).  The reference has no configuration precedent (SURVEY section 5 "config").
int DefaultDevice();

// sad.go:29-113.  One device worker replaces the goroutine pool; closing `in` stops it and closes
// `out`.  Chunks that are queued TOGETHER, belong to the same two images, span the same columns and follow each other
// downwards (the row bands every caller of the reference feeds: run.stream.go:46-58, sad.go:138-153 per tile row) are
// computed by ONE device call over their common rows and answered chunk by chunk, with one parameter snapshot for the
// group -- the reference's workers would each have taken theirs within the same microseconds (sad.go:51-53).  On a device error the worker fails `out` with the Error and keeps draining `in`, so a
// feeder blocked in send() is released and the consumer sees the error.
Pipeline SetupConcurrentSAD(int numWorkers, int device = -1);
// sad.go:172-202 (hole-free).  Throws the pipeline's Error -- or Error(SADHIP_ERR_ARG) for a channel
// closed early -- when fewer than `chunks` chunks arrive.
Gray AssembleDisparityMap(Chan<OutputChunk>& out, Rectangle dimensions, int chunks);
// sad.go:119-169: sets the global parameters, one frame-level device call.
Gray RunSad(const Gray& left, const Gray& right, int blockSize, int maxDisparity, int device = -1);
// The Go shim's RunSadFrame (host/go/sad_hip.go): RunSad WITHOUT the write to the process-wide parameters -- what a frame
// loop calls that only reads them (pkg/camera/output.go:169).  left and right travel with their own sizes.
Gray RunSadFrame(const Gray& left, const Gray& right, Parameters p, int device = -1);
// png.go:10-41: decode, then the reference's type switch -- *image.Gray is copied, everything else
// goes through the Go luminance formula on alpha-premultiplied 16-bit channels (gray.go:43-58),
// computed on the device for 8-bit RGB/NRGBA (sadhip_gray_u8).  Throws Error(SADHIP_ERR_ARG, ..)
// for unreadable files (LoadPNG returns (nil, err) in Go; MustLoadPNG panics).
struct LoadPNGOptions { bool refbug_rgba = false; int device = -1; };
Gray LoadPNG(const std::string& filename, LoadPNGOptions opt = {});
inline Gray MustLoadPNG(const std::string& filename) { return LoadPNG(filename); }
// png.go:54-66: 8-bit gray PNG, fastest compression level.
void SavePNG(const std::string& filename, const Gray& img);
inline void MustSavePNG(const std::string& filename, const Gray& img) { SavePNG(filename, img); }

// sad.go:205-244.
int SumAbsoluteDifferences(const Gray& left, const Gray& right, int leftX, int leftY, int rightX, int rightY, int blockSize);

// ---- in-memory handoff: the decoded samples of a PNG, as png.Decode left them ----
// kind/channels as in png_io.hpp (GoImage as int to keep that header out of this one).
struct Picture {
  int w = 0, h = 0, channels = 0, kind = 0; bool depth16 = false;
  std::vector<uint8_t> samples;
  bool device_gray_ok() const { return !depth16 && (channels == 3 || channels == 4); }   // 8-bit RGB / NRGBA
};
Picture LoadPicture(const std::string& filename);                   // decode only; nothing is converted or uploaded
Gray ToGray(const Picture& p, LoadPNGOptions opt = {});             // the type switch of png.go:30-37 on a decoded picture
// RunSad on decoded pictures.  Two 8-bit RGB(A) pictures of equal size and channel count: one
// sadhip_disparity_rgba_u8 call (uploads: one per image; the gray planes exist only in HBM).
// Anything else: ToGray + RunSad.  uploads_out (optional): number of host-to-device image uploads made.
Gray RunSad(const Picture& left, const Picture& right, int blockSize, int maxDisparity, int device = -1, int* uploads_out = nullptr);
// The Go shim's RunSadImages: the same without the write to the process-wide parameters -- what patches/0003 makes
// OutputCamera.processDepthMap call on the two decoded pictures instead of its color.GrayModel.Convert loops
// (pkg/camera/output.go:142-147,157-162).
Gray RunSadImages(const Picture& left, const Picture& right, Parameters p, int device = -1, int* uploads_out = nullptr);
// The Go shim's SetHostGray (libsadhip ABI 5, SADHIP_FLAG_HOST_GRAY): where RunSad / RunSadImages convert two colour pictures
// of one size to gray.  false (default): uploaded as they are, converted on the device; true: libsadhip's copy threads write
// luminance rows while they stage the pictures (the same integer formula, bit for bit) -- a quarter of the bytes over the bus.
void SetHostGray(bool on);
// device / wall time of the last RunSad of the calling thread ("elapsed" of output.go:200-204)
void LastTiming(float* kernel_ms, float* total_ms);

// ---- page-locked frames (sadhip_host_alloc): n frames of w x h u8, one contiguous block ------------
// What a capture pipeline fills instead of pageable Gray.Pix when it streams frames to the device: the
// host-buffer entry points recognise the memory and copy asynchronously at bus speed (include/sadhip.h).
class HostFrames {
 public:
  HostFrames(int n, int w, int h);
  ~HostFrames();
  HostFrames(const HostFrames&) = delete;
  HostFrames& operator=(const HostFrames&) = delete;
  int size() const { return n_; }
  int width() const { return w_; }
  int height() const { return h_; }
  uint8_t* frame(int i) { return base_ + (size_t)i * w_ * h_; }
  const uint8_t* frame(int i) const { return base_ + (size_t)i * w_ * h_; }
 private:
  uint8_t* base_ = nullptr; int n_ = 0, w_ = 0, h_ = 0;
};

// ---- several devices: frame i of a batch runs on devices[i mod G] --------------------
class DevicePool {
 public:
  DevicePool(const std::vector<int>& devices, int max_w, int max_h);
  ~DevicePool();
  DevicePool(const DevicePool&) = delete;
  DevicePool& operator=(const DevicePool&) = delete;
  int size() const;
  // one map per pair; the global parameters are set like RunSad does (sad.go:123-126)
  std::vector<Gray> RunSadBatch(const std::vector<std::shared_ptr<const Gray>>& lefts,
                                const std::vector<std::shared_ptr<const Gray>>& rights, int blockSize, int maxDisparity);
  // the same on page-locked frames, results into page-locked frames (all three of one size and length)
  void RunSadBatch(const HostFrames& lefts, const HostFrames& rights, HostFrames& outs, int blockSize, int maxDisparity);
 private:
  sadhip_pool* p_ = nullptr;
};

}  // namespace despair
