// host/copy_gang.h -- the host threads that move caller memory into / out of a context's page-locked planes (sadhip.hip).
// Plain C++17, no HIP: tests/test_copy_gang_cpu.py builds it alone under ThreadSanitizer and AddressSanitizer.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <emmintrin.h>
#include <mutex>
#include "gray_rows.h"
#include <thread>
#include <vector>

// The host threads that move caller memory into / out of the context's page-locked planes.
// One core copies frames that are not in its caches at 10-20 GB/s (tools/ab_hostcopy.py: 0.3-0.5 ms for
// the 6 MB of a 1080p pair), a fraction of what the bus then moves; the helpers split the rows with the
// calling thread.  The helpers touch host memory only -- no HIP call is made on them --
// are started on the first large copy of a context and inherit the affinity of the thread that
// starts them (a pool worker is pinned to the device's CPUs).
namespace sadhip {


struct RowCopy {
  uint8_t* dst; size_t dpitch; const uint8_t* src; size_t spitch; size_t wbytes;
  bool whole_rows;           // the bytes between the rows may be copied too (dst is a plane of ours, src readable to its last row's wbytes)
  // gray != 0: not a copy -- src rows are wpix pixels of `gray` (3 or 4) interleaved bytes (wbytes = wpix * gray, what is read),
  // dst rows receive wpix luminance bytes each (gray_rows.h: the Go formula), nothing beyond them
  int gray = 0, wpix = 0;
};

// memcpy with non-temporal stores for the 16-byte-aligned body of the destination (SSE2: every x86-64 has it).  A staged
// frame is written once and read next by a copy engine (upload) or never again by this thread (download): ordinary stores
// would first READ every destination line into the cache (a third of the memory traffic of the copy) and push the caller's
// working set out of it.  Pieces below kStreamMin go through memcpy.  copy_rows_* end with an sfence: the stores are
// globally visible before the thread reports its piece done.
constexpr size_t kStreamMin = 1024;
inline bool& stream_stores() { static bool on = true; return on; }      // (switched off by the diagnostics build for the A/B)

inline void copy_stream(uint8_t* dst, const uint8_t* src, size_t n)
{
  if (n < kStreamMin || !stream_stores()) { memcpy(dst, src, n); return; }
  const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
  size_t blocks = n / 64;
  for (; blocks; --blocks, dst += 64, src += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)(src + 16));
    const __m128i c = _mm_loadu_si128((const __m128i*)(src + 32)), d = _mm_loadu_si128((const __m128i*)(src + 48));
    _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)(dst + 16), b);
    _mm_stream_si128((__m128i*)(dst + 32), c); _mm_stream_si128((__m128i*)(dst + 48), d);
  }
  if (n & 63) memcpy(dst, src, n & 63);
}

inline void copy_part(const RowCopy& c, int r0, int r1)
{
  if (r1 <= r0) return;
  if (c.gray) {
    for (int r = r0; r < r1; ++r) gray_row(c.dst + (size_t)r * c.dpitch, c.src + (size_t)r * c.spitch, c.wpix, c.gray);
    _mm_sfence();
    return;
  }
  if (c.dpitch == c.spitch && (c.wbytes == c.dpitch || c.whole_rows)) {
    copy_stream(c.dst + (size_t)r0 * c.dpitch, c.src + (size_t)r0 * c.spitch, (size_t)(r1 - r0 - 1) * c.dpitch + c.wbytes);
  } else {
    for (int r = r0; r < r1; ++r) copy_stream(c.dst + (size_t)r * c.dpitch, c.src + (size_t)r * c.spitch, c.wbytes);
  }
  _mm_sfence();
}

class CopyGang {
 public:
  static constexpr size_t kMinBytes = (size_t)512 << 10;      // below this the calling thread copies alone
  static constexpr int kSpinUs = 150;   // default: a helper polls this long for the next piece before it sleeps (L, R and the result
                                        // of a call follow each other within that; waking a sleeping core costs 20-50 us per piece)
  explicit CopyGang(int helpers, int spin_us = kSpinUs) : helpers_(helpers), spin_us_(spin_us) {}
  // how long a thread polls before it sleeps on the condition variable; 0 = sleep at once (sadhip_ctx_set_copy_spin_us)
  void set_spin_us(int us) { spin_us_.store(us < 0 ? 0 : us, std::memory_order_relaxed); }
  int spin_us() const { return spin_us_.load(std::memory_order_relaxed); }
  ~CopyGang()
  {
    { std::lock_guard<std::mutex> lk(mu_); stop_.store(true, std::memory_order_release); }
    cv_work_.notify_all();
    for (auto& t : th_) t.join();
  }
  void rows(const RowCopy& c, int r0, int r1)
  {
    const int n = r1 - r0;
    if (helpers_ <= 0 || n < 2 * (helpers_ + 1) || (size_t)n * c.wbytes < kMinBytes || !start()) { copy_part(c, r0, r1); return; }
    const int parts = helpers_ + 1;
    {
      std::lock_guard<std::mutex> lk(mu_);                     // (the increment under the lock: no helper misses its wake-up)
      job_ = c; j0_ = r0; j1_ = r1;
      pending_.store(helpers_, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_work_.notify_all();
    copy_part(c, r0 + (int)((long long)n * helpers_ / parts), r1);          // the caller's own share: the last one
    if (!spin_until([&] { return pending_.load(std::memory_order_acquire) == 0; })) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
    }
  }

 private:
  template <class F> bool spin_until(F&& done) const
  {
    const int budget = spin_us_.load(std::memory_order_relaxed);
    if (budget <= 0) return done();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
      if (done()) return true;
      __builtin_ia32_pause();
      if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget)) return false;
    }
  }
  bool start()
  {
    if (!th_.empty()) return true;
    if (failed_) return false;
    try {
      for (int k = 0; k < helpers_; ++k) th_.emplace_back([this, k] { work(k); });
    } catch (...) {                                            // no threads to be had: the caller copies alone
      { std::lock_guard<std::mutex> lk(mu_); stop_.store(true, std::memory_order_release); }
      cv_work_.notify_all();
      for (auto& t : th_) t.join();
      th_.clear(); stop_.store(false, std::memory_order_release); failed_ = true;
      return false;
    }
    return true;
  }
  void work(int k)
  {
    unsigned seen = 0;
    for (;;) {
      auto news = [&] { return stop_.load(std::memory_order_acquire) || gen_.load(std::memory_order_acquire) != seen; };
      if (!spin_until(news)) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, news);
      }
      if (stop_.load(std::memory_order_acquire)) return;
      seen = gen_.load(std::memory_order_acquire);
      const RowCopy c = job_; const int r0 = j0_, r1 = j1_;   // stable until every helper has reported (rows() waits for that)
      const int n = r1 - r0, parts = helpers_ + 1;
      copy_part(c, r0 + (int)((long long)n * k / parts), r0 + (int)((long long)n * (k + 1) / parts));
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        { std::lock_guard<std::mutex> lk(mu_); }              // the caller is either polling or inside wait(): never in between
        cv_done_.notify_one();
      }
    }
  }
  const int helpers_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  RowCopy job_{}; int j0_ = 0, j1_ = 0;
  std::atomic<unsigned> gen_{0};
  std::atomic<int> pending_{0};
  std::atomic<bool> stop_{false};
  std::atomic<int> spin_us_;
  bool failed_ = false;
};

}  // namespace sadhip
