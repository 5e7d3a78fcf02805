// sadhip.hip -- C ABI (include/sadhip.h) over the gfx950 kernels.  Host side only:
// argument validation, per-device context (device buffers, pinned staging, streams),
// H2D / kernel / D2H pipelining.  No CPU compute fallback lives here.
#include "../../include/sadhip.h"
#include "sadhip_kernels.h"
#include "../host/copy_gang.h"

#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

using namespace sadhip;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int status, const char* fmt, ...)
{
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

#define HIPTRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(SADHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                  __FILE__, __LINE__);                                                    \
  } while (0)

extern "C" const char* sadhip_last_error(void) { return g_err; }

extern "C" const char* sadhip_status_string(int s)
{
  switch (s) {
    case SADHIP_OK: return "ok";
    case SADHIP_ERR_NULL: return "null pointer";
    case SADHIP_ERR_DIMS: return "bad image dimensions";
    case SADHIP_ERR_STRIDE: return "row stride smaller than width";
    case SADHIP_ERR_BLOCK: return "bad block size";
    case SADHIP_ERR_MAXDISP: return "bad max disparity";
    case SADHIP_ERR_REGION: return "bad region";
    case SADHIP_ERR_BATCH: return "bad batch size";
    case SADHIP_ERR_NO_DEVICE: return "no usable HIP device";
    case SADHIP_ERR_HIP: return "HIP runtime error";
    case SADHIP_ERR_ARG: return "invalid argument";
    case SADHIP_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}

extern "C" int sadhip_abi_version(void) { return SADHIP_ABI_VERSION; }

extern "C" int sadhip_device_count(int* count)
{
  if (!count) return fail(SADHIP_ERR_NULL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(SADHIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return SADHIP_OK;
}

// ------------------------------------------------------------------------------------------
// page-locked host memory this library knows about (sadhip_host_alloc / sadhip_host_register)
// ------------------------------------------------------------------------------------------
namespace {
struct PinnedRange { uintptr_t lo, hi; bool owned; };          // [lo, hi); owned: hipHostMalloc'ed here
std::shared_mutex g_pin_mu;
std::vector<PinnedRange> g_pins;                               // sorted by lo, disjoint

// [p, p + bytes) lies inside one known range
bool is_pinned(const void* p, size_t bytes)
{
  const uintptr_t a = (uintptr_t)p;
  std::shared_lock<std::shared_mutex> lk(g_pin_mu);
  auto it = std::upper_bound(g_pins.begin(), g_pins.end(), a, [](uintptr_t v, const PinnedRange& r) { return v < r.lo; });
  if (it == g_pins.begin()) return false;
  --it;
  return a >= it->lo && a + bytes <= it->hi;
}

// false: [p, p + bytes) overlaps a range already known (the registry stays sorted by lo and DISJOINT: is_pinned looks at
// one predecessor only, pin_remove identifies a range by its start)
bool pin_insert(void* p, size_t bytes, bool owned)
{
  const PinnedRange r{(uintptr_t)p, (uintptr_t)p + bytes, owned};
  std::unique_lock<std::shared_mutex> lk(g_pin_mu);
  auto it = std::upper_bound(g_pins.begin(), g_pins.end(), r.lo, [](uintptr_t v, const PinnedRange& q) { return v < q.lo; });
  if (it != g_pins.end() && it->lo < r.hi) return false;                    // the next range starts inside the new one
  if (it != g_pins.begin() && std::prev(it)->hi > r.lo) return false;       // the previous range reaches into it
  g_pins.insert(it, r);
  return true;
}

// [p, p + bytes) meets a known range -- at PAGE granularity: page-locking works on whole pages, and two registrations that
// share a page are two locks on it whose unlocks the driver has to keep apart (sadhip_host_register refuses that)
bool pin_overlaps(const void* p, size_t bytes)
{
  constexpr uintptr_t PG = 4096;
  const uintptr_t lo = (uintptr_t)p & ~(PG - 1), hi = ((uintptr_t)p + bytes + PG - 1) & ~(PG - 1);
  std::shared_lock<std::shared_mutex> lk(g_pin_mu);
  for (const PinnedRange& q : g_pins) {                       // (a handful of ranges; sorted by lo)
    const uintptr_t qlo = q.lo & ~(PG - 1), qhi = (q.hi + PG - 1) & ~(PG - 1);
    if (qlo < hi && lo < qhi) return true;
    if (qlo >= hi) break;
  }
  return false;
}

// removes the range that STARTS at p; false when there is none of that kind
bool pin_remove(void* p, bool owned)
{
  std::unique_lock<std::shared_mutex> lk(g_pin_mu);
  for (auto it = g_pins.begin(); it != g_pins.end(); ++it)
    if (it->lo == (uintptr_t)p && it->owned == owned) { g_pins.erase(it); return true; }
  return false;
}
}  // namespace

extern "C" int sadhip_host_alloc(size_t bytes, void** out)
{
  if (!out) return fail(SADHIP_ERR_NULL, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return fail(SADHIP_ERR_ARG, "bytes=0");
  void* p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);    // usable from every device (the pool)
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? SADHIP_ERR_NOMEM : (e == hipErrorNoDevice ? SADHIP_ERR_NO_DEVICE : SADHIP_ERR_HIP),
                "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  }
  bool fresh = false;
  try { fresh = pin_insert(p, bytes, true); }
  catch (const std::bad_alloc&) { (void)hipHostFree(p); return fail(SADHIP_ERR_NOMEM, "registry allocation failed"); }
  if (!fresh) {                                              // a stale sadhip_host_register entry covers memory the allocator reused
    (void)hipHostFree(p);
    return fail(SADHIP_ERR_ARG, "hipHostMalloc returned %p (+%zu), which overlaps a range still registered with sadhip_host_register", p, bytes);
  }
  *out = p;
  return SADHIP_OK;
}

extern "C" int sadhip_host_free(void* p)
{
  if (!p) return fail(SADHIP_ERR_NULL, "p is NULL");
  if (!pin_remove(p, true)) return fail(SADHIP_ERR_ARG, "%p was not returned by sadhip_host_alloc", p);
  HIPTRY(hipHostFree(p));
  return SADHIP_OK;
}

// What sadhip_host_register accepts (ABI 5).  hipHostRegister locks the CALLER's pages in place; the round-4 memory access
// faults happened on caller pages the runtime had locked in place, and both mechanisms the record admits
// -- a lock that outlives the memory it was taken on (the heap returns pages to the system and takes them back: brk / trim),
// or an access past the locked extent into a neighbouring heap object's page -- need the locked pages to be SHARED with, or
// recycled by, an allocator.  So a registration is whole pages of a mapping of the caller's own: 4 KiB-aligned start, a
// multiple of 4 KiB long, every page mapped readable + writable, and no part of it inside the program break ("[heap]": malloc's
// main arena, which grows and shrinks under the caller) or a stack.  Checked against /proc/self/maps; anything that cannot be
// verified is refused.  why: a sentence for sadhip_last_error().
static bool registrable(const void* p, size_t bytes, char* why, size_t cap)
{
  constexpr uintptr_t PG = 4096;
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  if ((lo & (PG - 1)) || (bytes & (PG - 1))) {
    snprintf(why, cap, "[%p, +%zu) is not whole 4 KiB pages (page-aligned start, length a multiple of 4096)", p, bytes);
    return false;
  }
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) { snprintf(why, cap, "/proc/self/maps is not readable: the mapping behind %p cannot be verified", p); return false; }
  char line[512];
  uintptr_t covered = lo;                                      // [lo, covered) lies in acceptable mappings so far
  bool ok = true;
  while (ok && covered < hi && fgets(line, sizeof(line), f)) {
    unsigned long a = 0, b = 0; char perms[8] = ""; int name_at = 0;
    if (sscanf(line, "%lx-%lx %7s %*s %*s %*s %n", &a, &b, perms, &name_at) < 3) continue;
    if (b <= covered) continue;                                // (maps are sorted by address)
    if (a > covered) break;                                    // a hole at `covered`
    const char* name = name_at > 0 ? line + name_at : "";
    if (perms[0] != 'r' || perms[1] != 'w') { snprintf(why, cap, "page %p of the range is not mapped readable and writable (%s)", (void*)covered, perms); ok = false; }
    else if (!strncmp(name, "[heap]", 6) || !strncmp(name, "[stack", 6)) {
      const int nlen = (int)strcspn(name, "]\n") + 1;                     // "[heap]" / "[stack]" without the line's end
      snprintf(why, cap, "[%p, +%zu) lies in %.*s: memory of an allocator that returns and re-takes pages under the caller cannot be page-locked in place -- use sadhip_host_alloc, or register a mapping of your own (mmap)", p, bytes, nlen, name);
      ok = false;
    }
    else covered = b;
  }
  fclose(f);
  if (ok && covered < hi) { snprintf(why, cap, "page %p of the range is not mapped", (void*)covered); ok = false; }
  return ok;
}

extern "C" int sadhip_host_register(void* p, size_t bytes)
{
  if (!p) return fail(SADHIP_ERR_NULL, "p is NULL");
  if (bytes == 0) return fail(SADHIP_ERR_ARG, "bytes=0");
  char why[400];
  if (!registrable(p, bytes, why, sizeof(why))) return fail(SADHIP_ERR_ARG, "%s", why);
  // ranges are disjoint: a sub-range of a sadhip_host_alloc block (already page-locked) or of an earlier registration is refused
  if (pin_overlaps(p, bytes)) return fail(SADHIP_ERR_ARG, "[%p, +%zu) shares a page with page-locked memory this library already knows", p, bytes);
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(e == hipErrorNoDevice ? SADHIP_ERR_NO_DEVICE : SADHIP_ERR_HIP, "hipHostRegister(%p, %zu): %s", p, bytes, hipGetErrorString(e));
  }
  bool fresh = false;
  try { fresh = pin_insert(p, bytes, false); }
  catch (const std::bad_alloc&) { (void)hipHostUnregister(p); return fail(SADHIP_ERR_NOMEM, "registry allocation failed"); }
  if (!fresh) {                                              // lost a race with another registration of overlapping memory
    (void)hipHostUnregister(p);
    return fail(SADHIP_ERR_ARG, "[%p, +%zu) overlaps page-locked memory this library already knows", p, bytes);
  }
  return SADHIP_OK;
}

extern "C" int sadhip_host_unregister(void* p)
{
  if (!p) return fail(SADHIP_ERR_NULL, "p is NULL");
  if (!pin_remove(p, false)) return fail(SADHIP_ERR_ARG, "%p was not registered with sadhip_host_register", p);
  HIPTRY(hipHostUnregister(p));
  return SADHIP_OK;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Slot {            // one pipeline stage worth of buffers (one frame)
  uint8_t *hL = nullptr, *hR = nullptr, *hO = nullptr;   // pinned host staging
  uint8_t *dL = nullptr, *dR = nullptr, *dO = nullptr;   // device planes
  hipStream_t stream = nullptr;
  // where the frame in flight runs: all three the slot's own stream for a single call; the context's per-direction streams
  // (s_up / s_kern / s_down) for a staged batch, chained by `uploaded` and k1 -- uploads, kernels and downloads of one stream
  // share a copy engine and serialise, and on some boxes of the pool three slot streams did not overlap at all (204 us per
  // pair = upload + kernel + download laid end to end, profiles/r04_ab_hostcopy_stream_stores.txt)
  hipStream_t up = nullptr, kern = nullptr, down = nullptr;
  hipEvent_t k0 = nullptr, k1 = nullptr;                 // kernel begin / end
  hipEvent_t uploaded = nullptr, done = nullptr;         // inputs on the device / result delivered (or in hO)
  static constexpr int kMaxBands = 4;
  hipEvent_t band[kMaxBands] = {};                       // band b of a staged result has arrived in hO
  uint8_t* out = nullptr; int ostride = 0;               // where the frame in flight delivers (route_output)
  bool out_direct = false;                               // ... straight into the caller's (page-locked) memory
  int out_bands = 1;
};

struct sadhip_ctx {
  int device = 0;
  int max_w = 0, max_h = 0, max_batch = 0;
  size_t pitch = 0;            // bytes per row of the staging planes
  size_t plane = 0;            // bytes per staging plane
#ifdef SADHIP_SLOTS
  static constexpr int kSlots = SADHIP_SLOTS;
#else
  static constexpr int kSlots = 3;
#endif
  Slot slot[kSlots];
  // Batches whose frames all lie in page-locked memory (sadhip_host_*) run through these instead: kGroup frames per
  // kernel launch, asynchronous linear copies, and ONE STREAM PER DIRECTION -- uploads, kernels and downloads each have
  // their own stream, chained by events per slot.  (Uploads and downloads issued on one stream share a DMA engine and
  // serialise: 36 GB/s host-to-device; a stream per direction reaches what tools/ubench/pcie_bw.hip measures for the
  // bus with both directions busy, 47 + 47 GB/s.)  Allocated on first use: kGroupSlots x 3 x kGroup planes.
  static constexpr int kGroup = 8, kGroupSlots = 4;
  struct GroupSlot {
    uint8_t *dL = nullptr, *dR = nullptr, *dO = nullptr;
    hipEvent_t up = nullptr, k0 = nullptr, k1 = nullptr, down = nullptr;     // uploaded / kernel begin, end / downloaded
  };
  GroupSlot grp[kGroupSlots];
  hipStream_t s_up = nullptr, s_kern = nullptr, s_down = nullptr;
  bool grp_ready = false;
  // scratch for sad_windows / gray (grown on demand)
  void* d_scratch = nullptr; size_t d_scratch_bytes = 0;
  void* h_scratch = nullptr; size_t h_scratch_bytes = 0;
  float last_kernel_ms = 0.f, last_total_ms = 0.f;
  // Host copies.  A caller buffer inside page-locked memory this library knows (sadhip_host_alloc / _register /
  // sadhip_pool_host_alloc) is handed to the copy engines as it is.  ANY OTHER host memory is copied by the host thread
  // into / out of the slot's own page-locked planes (hL, hR, hO), in bands so that the memcpy of one band overlaps the bus
  // transfer of the one before; no device-side copy ever touches memory the library did not lock itself.  Rounds 2-4
  // handed pageable caller memory to hipMemcpy2DAsync (the runtime locks the caller's pages in place: 0.14 ms instead of
  // 0.28 ms for the two uploads of a 1080p pair, tools/ubench/h2d_paths.hip); round 4 saw that path end a process with
  // "Memory access fault by GPU ... on address <a host heap page>" three times in some twenty full test runs (DESIGN.md
  // section 6b: profiles/r04_gpu_fault_{1,2}.log).  No build of the library contains that path any more (round 5); its A/B
  // numbers are kept in profiles/r04_ab_hostcopy*.txt.
  // Every host entry point synchronises its stream before returning -- on error exits too.
  // Staged copies are split two ways (tools/ab_hostcopy.py, profiles/r04_ab_hostcopy.txt; 1080p, BlockSize 9, MaxDisparity 64):
  // across kCopyHelpers + 1 host threads (CopyGang above) -- one thread: 0.39 ms per call on a pair that is in the caches,
  // 0.79 on 64 different pairs, 2 340 pairs/s batched; six threads: 0.27 / 0.30 ms, 7 600 pairs/s (the withdrawn in-place
  // path: 0.20 / 0.36 ms, 4 900 pairs/s) -- and into bands of about band_bytes, so that the bus moves band b while the
  // threads copy band b+1.  With the gang a band below 8 MB no longer pays (each costs a copy call and a gang hand-off:
  // sadhip_disparity_rgba_u8 0.55 ms at 8 MB, 0.63 at 2 MB): a 1080p plane or decoded picture goes in one piece, a 4K
  // RGBA picture in four.
  bool pair_upload = true;                     // a staged pair of a batch travels as one upload (diagnostics build: SADHIP_TUNE_PAIRUP=0)
  size_t band_bytes = (size_t)8 << 20;
  static constexpr int kCopyHelpers = SADHIP_DEFAULT_COPY_THREADS - 1;       // + the calling thread
  int copy_helpers = kCopyHelpers;
  int copy_spin_us = CopyGang::kSpinUs;
  std::unique_ptr<CopyGang> gang;
  std::mutex mu;
};

static int ensure_scratch(sadhip_ctx* c, size_t dbytes, size_t hbytes)
{
  if (dbytes > c->d_scratch_bytes) {
    if (c->d_scratch) HIPTRY(hipFree(c->d_scratch));
    c->d_scratch = nullptr; c->d_scratch_bytes = 0;
    HIPTRY(hipMalloc(&c->d_scratch, dbytes));
    c->d_scratch_bytes = dbytes;
  }
  if (hbytes > c->h_scratch_bytes) {
    if (c->h_scratch) HIPTRY(hipHostFree(c->h_scratch));
    c->h_scratch = nullptr; c->h_scratch_bytes = 0;
    HIPTRY(hipHostMalloc(&c->h_scratch, hbytes, hipHostMallocDefault));
    c->h_scratch_bytes = hbytes;
  }
  return SADHIP_OK;
}

static void destroy_groups(sadhip_ctx* c);

static void destroy_ctx(sadhip_ctx* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  destroy_groups(c);
  for (auto& s : c->slot) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.hL) (void)hipHostFree(s.hL);                       // (hR lives in the same block)
    if (s.hO) (void)hipHostFree(s.hO);
    if (s.dL) (void)hipFree(s.dL);                           // (dR lives in the same block)
    if (s.dO) (void)hipFree(s.dO);
    if (s.k0) (void)hipEventDestroy(s.k0);
    if (s.k1) (void)hipEventDestroy(s.k1);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.done) (void)hipEventDestroy(s.done);
    for (auto& b : s.band) if (b) (void)hipEventDestroy(b);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->h_scratch) (void)hipHostFree(c->h_scratch);
  delete c;
}

static void destroy_groups(sadhip_ctx* c)
{
  for (hipStream_t st : {c->s_up, c->s_kern, c->s_down}) if (st) (void)hipStreamSynchronize(st);
  for (auto& g : c->grp) {
    if (g.dL) (void)hipFree(g.dL);
    if (g.dR) (void)hipFree(g.dR);
    if (g.dO) (void)hipFree(g.dO);
    for (hipEvent_t e : {g.up, g.k0, g.k1, g.down}) if (e) (void)hipEventDestroy(e);
    g = sadhip_ctx::GroupSlot{};
  }
  for (hipStream_t* st : {&c->s_up, &c->s_kern, &c->s_down}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
  c->grp_ready = false;
}

// one stream per direction (uploads / kernels / downloads), shared by the page-locked group pipeline and the staged batch
static int ensure_dir_streams(sadhip_ctx* c)
{
  for (hipStream_t* st : {&c->s_up, &c->s_kern, &c->s_down})
    if (!*st) {
      const hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
      if (e != hipSuccess) { *st = nullptr; return fail(SADHIP_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e)); }
    }
  return SADHIP_OK;
}

static int ensure_groups(sadhip_ctx* c)
{
  if (c->grp_ready) return SADHIP_OK;
  if (int st = ensure_dir_streams(c)) return st;
  hipError_t e = hipSuccess;
  for (auto& g : c->grp) {
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g.up, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g.down, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&g.k0);
    if (e == hipSuccess) e = hipEventCreate(&g.k1);
    // + one row: the frames of a group keep the CALLER's strides, i.e. the last frame may end exactly at plane * kGroup.
    // No kernel reads a frame past byte (h-1)*stride + w (the column-wave kernels fetch rows with hardware range-checked
    // buffer loads whose record count is the row width, or with byte loads guarded by col < w; the generic kernels index
    // x < w only) -- the slack is insurance for that statement, not something a kernel relies on.
    const size_t gbytes = c->plane * sadhip_ctx::kGroup + c->pitch;
    if (e == hipSuccess) e = hipMalloc((void**)&g.dL, gbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&g.dR, gbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&g.dO, gbytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    destroy_groups(c);
    return fail(e == hipErrorOutOfMemory ? SADHIP_ERR_NOMEM : SADHIP_ERR_HIP, "group buffers (%d x %d x %zu B): %s",
                sadhip_ctx::kGroupSlots, sadhip_ctx::kGroup, c->plane, hipGetErrorString(e));
  }
  c->grp_ready = true;
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_create(int device, int max_w, int max_h, int max_batch, sadhip_ctx** out)
{
  if (!out) return fail(SADHIP_ERR_NULL, "out is NULL");
  *out = nullptr;
  if (max_w <= 0 || max_h <= 0) return fail(SADHIP_ERR_DIMS, "max_w=%d max_h=%d", max_w, max_h);
  if (max_batch < 1) return fail(SADHIP_ERR_BATCH, "max_batch=%d", max_batch);   // a hint only: batches of any length stream through two frame slots
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SADHIP_ERR_NO_DEVICE, "no HIP device");
  if (device < 0 || device >= n) return fail(SADHIP_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
  HIPTRY(hipSetDevice(device));

  sadhip_ctx* c = new (std::nothrow) sadhip_ctx();
  if (!c) return fail(SADHIP_ERR_NOMEM, "context allocation failed");
  c->device = device; c->max_w = max_w; c->max_h = max_h; c->max_batch = max_batch;
  c->pitch = align_up((size_t)max_w, 256);
  c->plane = c->pitch * (size_t)max_h;
#ifdef SADHIP_TUNING
  if (const char* e = getenv("SADHIP_TUNE_STREAM")) sadhip::stream_stores() = atoi(e) != 0;
  if (const char* e = getenv("SADHIP_TUNE_PAIRUP")) c->pair_upload = atoi(e) != 0;
  if (const char* e = getenv("SADHIP_TUNE_BANDKB")) c->band_bytes = (size_t)std::max(atoi(e), 4) << 10;
  int helpers = sadhip_ctx::kCopyHelpers;
  if (const char* e = getenv("SADHIP_TUNE_COPYTHREADS")) helpers = std::min(std::max(atoi(e), 1), 32) - 1;
  c->copy_helpers = helpers;
  c->gang.reset(new (std::nothrow) CopyGang(helpers));
#else
  c->gang.reset(new (std::nothrow) CopyGang(sadhip_ctx::kCopyHelpers));
#endif
  for (auto& s : c->slot) {
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&s.k0);
    if (e == hipSuccess) e = hipEventCreate(&s.k1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    s.up = s.kern = s.down = s.stream;
    for (auto& b : s.band) if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&s.dL, 2 * c->plane);      // L and R planes in one block: a staged pair can
    if (e == hipSuccess) s.dR = s.dL + c->plane;                          // travel as ONE upload (enqueue_frame)
    if (e == hipSuccess) e = hipMalloc((void**)&s.dO, c->plane);
    if (e != hipSuccess) {
      int st = fail(e == hipErrorOutOfMemory ? SADHIP_ERR_NOMEM : SADHIP_ERR_HIP,
                    "context buffers (%zu B planes): %s", c->plane, hipGetErrorString(e));
      destroy_ctx(c);
      return st;
    }
  }
  *out = c;
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_destroy(sadhip_ctx* ctx)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  destroy_ctx(ctx);
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_set_copy_threads(sadhip_ctx* ctx, int n)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  if (n < 1 || n > 32) return fail(SADHIP_ERR_ARG, "copy threads n=%d (valid 1..32, the calling thread included)", n);
  std::lock_guard<std::mutex> lk(ctx->mu);                  // no call is inside a copy
  if (n - 1 == ctx->copy_helpers && ctx->gang) return SADHIP_OK;
  CopyGang* g = new (std::nothrow) CopyGang(n - 1, ctx->copy_spin_us);
  if (!g) return fail(SADHIP_ERR_NOMEM, "copy threads");
  ctx->gang.reset(g);                                       // joins the old helpers
  ctx->copy_helpers = n - 1;
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_get_copy_threads(sadhip_ctx* ctx, int* n)
{
  if (!ctx || !n) return fail(SADHIP_ERR_NULL, "ctx or n is NULL");
  std::lock_guard<std::mutex> lk(ctx->mu);
  *n = ctx->copy_helpers + 1;
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_set_copy_spin_us(sadhip_ctx* ctx, int us)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  if (us < 0 || us > SADHIP_MAX_COPY_SPIN_US) return fail(SADHIP_ERR_ARG, "copy spin us=%d (valid 0..%d)", us, SADHIP_MAX_COPY_SPIN_US);
  std::lock_guard<std::mutex> lk(ctx->mu);                  // no call is inside a copy
  ctx->copy_spin_us = us;
  if (ctx->gang) ctx->gang->set_spin_us(us);
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_get_copy_spin_us(sadhip_ctx* ctx, int* us)
{
  if (!ctx || !us) return fail(SADHIP_ERR_NULL, "ctx or us is NULL");
  std::lock_guard<std::mutex> lk(ctx->mu);
  *us = ctx->copy_spin_us;
  return SADHIP_OK;
}

extern "C" int sadhip_ctx_device(const sadhip_ctx* ctx, int* device)
{
  if (!ctx || !device) return fail(SADHIP_ERR_NULL, "ctx or device is NULL");
  *device = ctx->device;
  return SADHIP_OK;
}

// ------------------------------------------------------------------------------------------
// validation shared by every disparity entry point
// ------------------------------------------------------------------------------------------
static int check_params(int w, int h, int lstride, int rstride, int block, int maxdisp)
{
  if (w <= 0 || h <= 0) return fail(SADHIP_ERR_DIMS, "w=%d h=%d", w, h);
  if (lstride < w || rstride < w) return fail(SADHIP_ERR_STRIDE, "lstride=%d rstride=%d w=%d", lstride, rstride, w);
  if (block < 1 || block > SADHIP_MAX_BLOCK) return fail(SADHIP_ERR_BLOCK, "block=%d (valid 1..%d)", block, SADHIP_MAX_BLOCK);
  if (maxdisp < 1 || maxdisp > SADHIP_MAX_MAXDISP)
    return fail(SADHIP_ERR_MAXDISP, "maxdisp=%d (valid 1..%d; the reference divides by MaxDisparity, sad.go:92)", maxdisp, SADHIP_MAX_MAXDISP);
  return SADHIP_OK;
}

// extra: flag bits only the calling entry point knows (SADHIP_FLAG_HOST_GRAY: sadhip_disparity_rgba_u8)
static int check_flags(unsigned flags, unsigned extra = 0)
{
  const unsigned mirrors = SADHIP_FLAG_MIRROR_SADC | SADHIP_FLAG_MIRROR_TESTPY | SADHIP_FLAG_MIRROR_VERILOG;
  const unsigned known = SADHIP_FLAG_RAW | SADHIP_FLAG_FORCE_GENERIC | mirrors | extra;
  if (flags & ~known) return fail(SADHIP_ERR_ARG, "unknown flag bits 0x%x", flags & ~known);
  const unsigned m = flags & mirrors;
  if (m & (m - 1)) return fail(SADHIP_ERR_ARG, "flags 0x%x select more than one hardware-mirror mode", flags);
  return SADHIP_OK;
}

static hipError_t dispatch(const Job& j, unsigned flags, hipStream_t s)
{
  if (flags & SADHIP_FLAG_MIRROR_SADC) return launch_mirror(j, 1, s);
  if (flags & SADHIP_FLAG_MIRROR_TESTPY) return launch_mirror(j, 2, s);
  if (flags & SADHIP_FLAG_MIRROR_VERILOG) return launch_mirror(j, 3, s);
  // launch_fast plans the job once and routes it to the generic kernel itself when the column-wave kernel has no
  // plan for it (sadhip_kernel_kind() gives the same answer up front)
  if (!(flags & SADHIP_FLAG_FORCE_GENERIC)) return launch_fast(j, s);
  return launch_generic(j, s);
}

extern "C" int sadhip_plan_describe(int w, int h, int block, int maxdisp, char* buf, int cap)
{
  if (buf && cap > 0) buf[0] = 0;
  if (check_params(w, h, w, w, block, maxdisp) != SADHIP_OK) return -1;
  Job j{}; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp; j.n = 1;
  j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.lstride = j.rstride = j.ostride = w;
  return describe_fast(j, buf, cap, false);
}

extern "C" int sadhip_plan_describe_batch(int w, int h, int block, int maxdisp, int n, char* buf, int cap)
{
  if (buf && cap > 0) buf[0] = 0;
  if (n < 1 || check_params(w, h, w, w, block, maxdisp) != SADHIP_OK) return -1;
  Job j{}; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp; j.n = n;
  j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.lstride = j.rstride = j.ostride = w;
  return describe_fast(j, buf, cap, true);
}

extern "C" int sadhip_last_kernel_kind(void) { return last_launch_kind(); }

extern "C" int sadhip_kernel_kind(int w, int h, int block, int maxdisp, unsigned flags)
{
  if (check_params(w, h, w, w, block, maxdisp) != SADHIP_OK || check_flags(flags) != SADHIP_OK) return -1;
  if (flags & SADHIP_FLAG_MIRROR_SADC) return KIND_MIRROR_SADC;
  if (flags & SADHIP_FLAG_MIRROR_TESTPY) return KIND_MIRROR_TESTPY;
  if (flags & SADHIP_FLAG_MIRROR_VERILOG) return KIND_MIRROR_VERILOG;
  Job j{}; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp; j.n = 1;
  j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.lstride = j.rstride = j.ostride = w;
  const int generic = generic_uses_sliding(block, maxdisp) ? KIND_GENERIC_SLIDING : KIND_GENERIC;
  if (flags & SADHIP_FLAG_FORCE_GENERIC) return generic;
  return fast_supported(j) ? KIND_COLWAVE : generic;
}

using clk = std::chrono::steady_clock;

// row pitch of the staging planes for a w-wide frame (<= the context's allocation pitch)
static inline size_t pitch_for(int w) { return align_up((size_t)w, 256); }

// rows of `pitch` bytes cut into bands of whole rows, about band_bytes each (at most Slot::kMaxBands, none shorter than 64 rows)
static inline int band_count(int rows, size_t pitch, size_t band_bytes)
{
  const size_t want = ((size_t)rows * pitch + band_bytes / 2) / band_bytes;
  return (int)std::max<size_t>(1, std::min<size_t>({want, (size_t)Slot::kMaxBands, (size_t)rows / 64}));
}
static inline int band_row(int r0, int r1, int nb, int b) { return r0 + (int)((long long)(r1 - r0) * b / nb); }

// the slot's page-locked host planes: allocated when the first frame that needs staging arrives (a context that only ever
// sees device-resident or page-locked frames never locks host memory of its own)
static int ensure_stage(sadhip_ctx* c, Slot& s)
{
  struct { uint8_t** p; size_t bytes; } want[] = {{&s.hL, 2 * c->plane}, {&s.hO, c->plane}};     // hL + hR in one block, like dL + dR
  for (auto& w : want)
    if (!*w.p) {
      const hipError_t e = hipHostMalloc((void**)w.p, w.bytes, hipHostMallocDefault);
      if (e != hipSuccess) {
        *w.p = nullptr;
        return fail(e == hipErrorOutOfMemory ? SADHIP_ERR_NOMEM : SADHIP_ERR_HIP, "page-locked staging plane (%zu B): %s", w.bytes, hipGetErrorString(e));
      }
    }
  s.hR = s.hL + c->plane;
  return SADHIP_OK;
}

static inline void copy_rows(sadhip_ctx* c, const RowCopy& rc, int rows)
{
  if (c->gang) c->gang->rows(rc, 0, rows); else copy_part(rc, 0, rows);
}

// Rows [r0, r1) of a host image that is NOT in page-locked memory this library knows -> device rows of `pitch` bytes
// (device row 0 = image row r0), through the page-locked plane `stage`: the host thread copies a band, queues its transfer
// and copies the next band while the first is on the bus.
static int upload_staged(sadhip_ctx* c, hipStream_t st, uint8_t* dev, uint8_t* stage, size_t pitch,
                         const uint8_t* src, size_t stride, size_t wbytes, int r0, int r1)
{
  if (r1 <= r0) return SADHIP_OK;                           // (a right picture that ends above the band: nothing of it exists here)
  const int nb = band_count(r1 - r0, pitch, c->band_bytes);
  for (int b = 0; b < nb; ++b) {
    const int a0 = band_row(r0, r1, nb, b), a1 = band_row(r0, r1, nb, b + 1);
    if (a1 <= a0) continue;
    uint8_t* hb = stage + (size_t)(a0 - r0) * pitch;
    // stage row 0 of the band = image row a0; never a byte past an image row's wbytes on the last row of a part
    copy_rows(c, RowCopy{hb, pitch, src + (size_t)a0 * stride, stride, wbytes, true}, a1 - a0);
    HIPTRY(hipMemcpyAsync(dev + (size_t)(a0 - r0) * pitch, hb, (size_t)(a1 - a0) * pitch, hipMemcpyHostToDevice, st));
  }
  return SADHIP_OK;
}

// Where the result of the frame about to be queued on slot s goes: straight into the caller's rows when they lie in
// page-locked memory this library knows, through the slot's page-locked plane otherwise.  Call before enqueue_frame.
static void route_output(sadhip_ctx* c, Slot& s, uint8_t* out, int ostride, int wbytes, int rows)
{
  s.out = out; s.ostride = ostride;
  (void)c;
  s.out_direct = is_pinned(out, (size_t)(rows - 1) * ostride + wbytes);
}

static int enqueue_download(sadhip_ctx* c, Slot& s, int w, int x0, int x1, int rows);

// One frame (or band of a frame) through slot s: upload -> kernel -> download, all async on the slot's stream.  Rows
// [band0,band1) of the inputs are uploaded.
// rw, rh: the part of the right image that exists (<= w, h; == w, h unless the caller's right image is the smaller
// one, sadhip_disparity_lr_u8): only that part is read from the caller's memory.
static int enqueue_frame(sadhip_ctx* c, Slot& s, const uint8_t* left, const uint8_t* right,
                         int w, int h, int lstride, int rstride, int block, int maxdisp, unsigned flags,
                         int x0, int y0, int x1, int y1, int band0, int band1, int rw = 0, int rh = 0)
{
  const size_t pitch = pitch_for(w);
  if (rw <= 0 || rw > w) rw = w;
  if (rh <= 0 || rh > h) rh = h;
  const int rband1 = band1 < rh ? band1 : rh;             // right rows [band0, rband1) exist
  // Page-locked inputs this library knows (sadhip_host_*): one asynchronous linear copy per image, the device rows
  // keep the caller's stride (nothing is staged and the host thread does not wait for the bus).
  size_t dl = pitch, dr = pitch;                           // device row strides
  const uint8_t* dright = s.dR;                            // where the right rows land on the device
  const size_t lbytes = (size_t)(band1 - band0 - 1) * lstride + w;
  const size_t rbytes = rband1 > band0 ? (size_t)(rband1 - band0 - 1) * rstride + rw : 0;
  const bool pinned = (size_t)lstride * (band1 - band0) <= c->plane && (size_t)rstride * (band1 - band0) <= c->plane &&
                      is_pinned(left + (size_t)band0 * lstride, lbytes) && (rbytes == 0 || is_pinned(right + (size_t)band0 * rstride, rbytes));
  if (pinned) {
    dl = (size_t)lstride; dr = (size_t)rstride;
    HIPTRY(hipMemcpyAsync(s.dL, left + (size_t)band0 * lstride, lbytes, hipMemcpyHostToDevice, s.up));
    if (rbytes) HIPTRY(hipMemcpyAsync(s.dR, right + (size_t)band0 * rstride, rbytes, hipMemcpyHostToDevice, s.up));
  } else {
    if (int st = ensure_stage(c, s)) return st;
    const int lrows = band1 - band0, rrows = rband1 - band0;
    // (round 5) ... unless the pair is small: below 1 MiB for both planes the second copy's fixed cost (~10 us) is more than the
    // overlap of the left plane's transfer with the right plane's staging is worth -- the reference's own frames are 640x480
    const bool small_pair = (size_t)(lrows + rrows) * pitch <= ((size_t)1 << 20);
    if ((s.up != s.stream || small_pair) && c->pair_upload && rrows > 0 && band_count(lrows, pitch, c->band_bytes) == 1 &&
        band_count(rrows, pitch, c->band_bytes) == 1) {
      // a frame of a batch: the right rows right behind the left ones, the pair as ONE upload -- a 2.2 MB copy costs 54 us of
      // which ~10 are fixed, and two of them per pair were the upload stream's limit (9 100 pairs/s at 1080p).  Not for a
      // single call: there the left plane's transfer overlaps the staging of the right one.
      const size_t used = (size_t)lrows * pitch;
      copy_rows(c, RowCopy{s.hL, pitch, left + (size_t)band0 * lstride, (size_t)lstride, (size_t)w, true}, lrows);
      copy_rows(c, RowCopy{s.hL + used, pitch, right + (size_t)band0 * rstride, (size_t)rstride, (size_t)rw, true}, rrows);
      HIPTRY(hipMemcpyAsync(s.dL, s.hL, used + (size_t)rrows * pitch, hipMemcpyHostToDevice, s.up));
      dright = s.dL + used;
    } else {
      if (int st = upload_staged(c, s.up, s.dL, s.hL, pitch, left, (size_t)lstride, (size_t)w, band0, band1)) return st;
      if (int st = upload_staged(c, s.up, s.dR, s.hR, pitch, right, (size_t)rstride, (size_t)rw, band0, rband1)) return st;
    }
  }
  Job j{};
  j.left = s.dL; j.right = dright; j.out = s.dO;
  j.left_frame_stride = j.right_frame_stride = j.out_frame_stride = c->plane;
  j.lstride = (int)dl; j.rstride = (int)dr; j.ostride = (int)pitch;
  j.n = 1; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp;
  j.rw = rw < w ? rw : 0; j.rh = rh < h ? rh : 0;
  j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
  j.ox = 0; j.oy = y0;            // staging output: row 0 = image row y0, full-width rows
  j.raw = (flags & SADHIP_FLAG_RAW) ? 1 : 0;
  j.row0 = band0;
  if (s.kern != s.up) {
    HIPTRY(hipEventRecord(s.uploaded, s.up));
    HIPTRY(hipStreamWaitEvent(s.kern, s.uploaded, 0));
  }
  HIPTRY(hipEventRecord(s.k0, s.kern));
  HIPTRY(dispatch(j, flags, s.kern));
  HIPTRY(hipEventRecord(s.k1, s.kern));
  return enqueue_download(c, s, w, x0, x1, y1 - y0);
}

// queue the download of result rows [0, rows) of slot s (after its kernel): into the caller's page-locked rows, or into
// the slot's page-locked plane band by band (finish_frame copies each band out as it arrives)
static int enqueue_download(sadhip_ctx* c, Slot& s, int w, int x0, int x1, int rows)
{
  const size_t pitch = pitch_for(w);
  if (s.down != s.kern) HIPTRY(hipStreamWaitEvent(s.down, s.k1, 0));
  if (s.out_direct) {
    HIPTRY(hipMemcpy2DAsync(s.out, s.ostride, s.dO + x0, pitch, x1 - x0, rows, hipMemcpyDeviceToHost, s.down));
    HIPTRY(hipEventRecord(s.done, s.down));
    return SADHIP_OK;
  }
  if (int st = ensure_stage(c, s)) return st;
  s.out_bands = band_count(rows, pitch, c->band_bytes);
  for (int b = 0; b < s.out_bands; ++b) {
    const int a0 = band_row(0, rows, s.out_bands, b), a1 = band_row(0, rows, s.out_bands, b + 1);
    HIPTRY(hipMemcpyAsync(s.hO + (size_t)a0 * pitch, s.dO + (size_t)a0 * pitch, (size_t)(a1 - a0) * pitch, hipMemcpyDeviceToHost, s.down));
    HIPTRY(hipEventRecord(s.band[b], s.down));
  }
  HIPTRY(hipEventRecord(s.done, s.down));
  return SADHIP_OK;
}

// wait for the frame queued on slot s and hand a staged result to the caller's rows (s.out, route_output)
static int finish_frame(sadhip_ctx* c, Slot& s, int w, int x0, int y0, int x1, int y1, float* kernel_ms)
{
  const size_t pitch = pitch_for(w);
  const int rows = y1 - y0, rw = x1 - x0;
  uint8_t* out = s.out; const int ostride = s.ostride;
  if (!s.out_direct)
    for (int b = 0; b < s.out_bands; ++b) {
      HIPTRY(hipEventSynchronize(s.band[b]));
      const int a0 = band_row(0, rows, s.out_bands, b), a1 = band_row(0, rows, s.out_bands, b + 1);
      // never a byte outside the caller's rw-byte rows
      copy_rows(c, RowCopy{out + (size_t)a0 * ostride, (size_t)ostride, s.hO + (size_t)a0 * pitch + x0, pitch, (size_t)rw, false}, a1 - a0);
    }
  HIPTRY(hipEventSynchronize(s.done));                     // (after the kernel's end on whichever stream ran it)
  float ms = 0.f;
  HIPTRY(hipEventElapsedTime(&ms, s.k0, s.k1));
  if (kernel_ms) *kernel_ms += ms;
  return SADHIP_OK;
}

extern "C" int sadhip_disparity_region_u8(sadhip_ctx* ctx, const uint8_t* left, const uint8_t* right,
                                          int w, int h, int lstride, int rstride,
                                          int block, int maxdisp, unsigned flags,
                                          int x0, int y0, int x1, int y1, uint8_t* out_region)
{
  if (!ctx || !left || !right || !out_region) return fail(SADHIP_ERR_NULL, "ctx/left/right/out is NULL");
  if (int st = check_params(w, h, lstride, rstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (w > ctx->max_w || h > ctx->max_h) return fail(SADHIP_ERR_DIMS, "%dx%d exceeds context capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
  if (x0 < 0 || y0 < 0 || x1 > w || y1 > h || x0 >= x1 || y0 >= y1)
    return fail(SADHIP_ERR_REGION, "region (%d,%d)-(%d,%d) not inside %dx%d", x0, y0, x1, y1, w, h);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const auto t0 = clk::now();
  const int hs = block / 2;
  const int band0 = y0 - hs < 0 ? 0 : y0 - hs;
  const int band1 = y1 + hs > h ? h : y1 + hs;
  Slot& s = ctx->slot[0];
  float kms = 0.f;
  route_output(ctx, s, out_region, x1 - x0, x1 - x0, y1 - y0);
  int st = enqueue_frame(ctx, s, left, right, w, h, lstride, rstride, block, maxdisp, flags, x0, y0, x1, y1, band0, band1);
  if (st == SADHIP_OK) st = finish_frame(ctx, s, w, x0, y0, x1, y1, &kms);
  if (st != SADHIP_OK) { (void)hipStreamSynchronize(s.stream); return st; }   // no copy from/to caller memory left in flight
  ctx->last_kernel_ms = kms;
  ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return SADHIP_OK;
}

// Left and right images with their own sizes (pkg/camera/output.go decodes the two PNGs independently).
// A right image at least as large as the left one is the ordinary job on its top-left lw x lh part (nothing
// beyond it is ever addressed: right indices are <= left indices in both axes).  A smaller one drops the window
// terms whose right pixel does not exist (sad.go:225-227,231-233): generic kernel, see Job::rw.
extern "C" int sadhip_disparity_lr_u8(sadhip_ctx* ctx,
                                      const uint8_t* left, int lw, int lh, int lstride,
                                      const uint8_t* right, int rw, int rh, int rstride,
                                      int block, int maxdisp, unsigned flags,
                                      int x0, int y0, int x1, int y1, uint8_t* out, int ostride)
{
  if (!ctx || !left || !right || !out) return fail(SADHIP_ERR_NULL, "ctx/left/right/out is NULL");
  if (rw <= 0 || rh <= 0) return fail(SADHIP_ERR_DIMS, "right image %dx%d", rw, rh);
  if (rstride < rw) return fail(SADHIP_ERR_STRIDE, "rstride=%d rw=%d", rstride, rw);
  if (int st = check_params(lw, lh, lstride, lstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (lw > ctx->max_w || lh > ctx->max_h) return fail(SADHIP_ERR_DIMS, "%dx%d exceeds context capacity %dx%d", lw, lh, ctx->max_w, ctx->max_h);
  if (x0 < 0 || y0 < 0 || x1 > lw || y1 > lh || x0 >= x1 || y0 >= y1)
    return fail(SADHIP_ERR_REGION, "region (%d,%d)-(%d,%d) not inside %dx%d", x0, y0, x1, y1, lw, lh);
  if (ostride < x1 - x0) return fail(SADHIP_ERR_STRIDE, "ostride=%d region width=%d", ostride, x1 - x0);
  const bool smaller = rw < lw || rh < lh;
  const unsigned mirrors = SADHIP_FLAG_MIRROR_SADC | SADHIP_FLAG_MIRROR_TESTPY | SADHIP_FLAG_MIRROR_VERILOG;
  if (smaller && (flags & mirrors))
    return fail(SADHIP_ERR_DIMS, "hardware-mirror modes need a right image of at least the left image's size (%dx%d < %dx%d)", rw, rh, lw, lh);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const auto t0 = clk::now();
  const int hs = block / 2;
  const int band0 = y0 - hs < 0 ? 0 : y0 - hs;
  const int band1 = y1 + hs > lh ? lh : y1 + hs;
  Slot& s = ctx->slot[0];
  float kms = 0.f;
  route_output(ctx, s, out, ostride, x1 - x0, y1 - y0);
  int st = enqueue_frame(ctx, s, left, right, lw, lh, lstride, rstride, block, maxdisp, flags, x0, y0, x1, y1, band0, band1, rw, rh);
  if (st == SADHIP_OK) st = finish_frame(ctx, s, lw, x0, y0, x1, y1, &kms);
  if (st != SADHIP_OK) { (void)hipStreamSynchronize(s.stream); return st; }
  ctx->last_kernel_ms = kms;
  ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return SADHIP_OK;
}

extern "C" int sadhip_disparity_u8(sadhip_ctx* ctx, const uint8_t* left, const uint8_t* right,
                                   int w, int h, int lstride, int rstride,
                                   int block, int maxdisp, unsigned flags, uint8_t* out, int ostride)
{
  if (!ctx || !left || !right || !out) return fail(SADHIP_ERR_NULL, "ctx/left/right/out is NULL");
  if (int st = check_params(w, h, lstride, rstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (ostride < w) return fail(SADHIP_ERR_STRIDE, "ostride=%d w=%d", ostride, w);
  if (w > ctx->max_w || h > ctx->max_h) return fail(SADHIP_ERR_DIMS, "%dx%d exceeds context capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const auto t0 = clk::now();
  Slot& s = ctx->slot[0];
  float kms = 0.f;
  route_output(ctx, s, out, ostride, w, h);
  int st = enqueue_frame(ctx, s, left, right, w, h, lstride, rstride, block, maxdisp, flags, 0, 0, w, h, 0, h);
  if (st == SADHIP_OK) st = finish_frame(ctx, s, w, 0, 0, w, h, &kms);
  if (st != SADHIP_OK) { (void)hipStreamSynchronize(s.stream); return st; }
  ctx->last_kernel_ms = kms;
  ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return SADHIP_OK;
}

// A batch whose frames all lie in page-locked memory: groups of kGroup frames, one kernel launch per group, linear
// asynchronous copies (frames that follow each other in the caller's memory as one copy).  Three streams -- uploads,
// kernels, downloads -- and per slot the events that chain them:
//   upload g    waits for kernel g-S (it read the slot's input planes)          -> records up
//   kernel g    waits for up, and for download g-S (it read the slot's output)  -> records k0 .. k1
//   download g  waits for k1                                                    -> records down
// The host runs at most S groups ahead (it waits for download g-S before it reuses a slot) and drains all three
// streams before it returns -- on errors too.  The caller holds ctx->mu and has validated everything.
static int batch_pinned(sadhip_ctx* c, int n, const uint8_t* const* lefts, const uint8_t* const* rights,
                        int w, int h, int lstride, int rstride, int block, int maxdisp, unsigned flags,
                        uint8_t* const* outs, int ostride, float* kms)
{
  if (int st = ensure_groups(c)) return st;
  constexpr int G = sadhip_ctx::kGroup, S = sadhip_ctx::kGroupSlots;
  const size_t fl = (size_t)lstride * h, fr = (size_t)rstride * h, fo = (size_t)ostride * h;      // frame to frame
  const size_t cl = fl - lstride + w, cr = fr - rstride + w, co = fo - ostride + w;               // bytes of one frame
  const int ng = (n + G - 1) / G;
  // copies of frames [i0, i0 + cnt): a run of frames that follow each other at exactly `step` bytes inside ONE
  // page-locked range travels as one copy
  auto copy_runs = [&](hipStream_t st, uint8_t* dev, const uint8_t* const* host, int i0, int cnt, size_t step, size_t one, bool to_dev) -> int {
    for (int i = 0; i < cnt;) {
      int k = i + 1;
      while (k < cnt && host[i0 + k] == host[i0 + k - 1] + step) ++k;
      if (k - i > 1 && !is_pinned(host[i0 + i], (size_t)(k - i - 1) * step + one)) k = i + 1;
      const size_t bytes = (size_t)(k - i - 1) * step + one;
      if (to_dev) HIPTRY(hipMemcpyAsync(dev + (size_t)i * step, host[i0 + i], bytes, hipMemcpyHostToDevice, st));
      else HIPTRY(hipMemcpyAsync(const_cast<uint8_t*>(host[i0 + i]), dev + (size_t)i * step, bytes, hipMemcpyDeviceToHost, st));
      i = k;
    }
    return SADHIP_OK;
  };
  // Results: a linear copy also carries the bytes BETWEEN rows, which on the way down would overwrite whatever the
  // caller keeps there (an output that is a column window of a wider image: ostride > w) -- linear only for tight
  // rows, a 2-D copy per frame otherwise.  (On the way up the extra bytes are merely read, inside a verified range.)
  auto download = [&](hipStream_t st, const uint8_t* dev, int i0, int cnt) -> int {
    if (ostride == w) return copy_runs(st, const_cast<uint8_t*>(dev), outs, i0, cnt, fo, co, false);
    for (int i = 0; i < cnt; ++i)
      HIPTRY(hipMemcpy2DAsync(outs[i0 + i], (size_t)ostride, dev + (size_t)i * fo, (size_t)ostride, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, st));
    return SADHIP_OK;
  };
  auto body = [&]() -> int {
    for (int g = 0; g < ng; ++g) {
      sadhip_ctx::GroupSlot& s = c->grp[g % S];
      const int i0 = g * G, cnt = n - i0 < G ? n - i0 : G;
      if (g >= S) {                                             // the slot's previous group: results home, kernel time read
        HIPTRY(hipEventSynchronize(s.down));
        float ms = 0.f;
        HIPTRY(hipEventElapsedTime(&ms, s.k0, s.k1));
        *kms += ms;
        HIPTRY(hipStreamWaitEvent(c->s_up, s.k1, 0));          // (complete already: the download waited for it)
      }
      if (int st = copy_runs(c->s_up, s.dL, lefts, i0, cnt, fl, cl, true)) return st;
      if (int st = copy_runs(c->s_up, s.dR, rights, i0, cnt, fr, cr, true)) return st;
      HIPTRY(hipEventRecord(s.up, c->s_up));
      Job j{};
      j.left = s.dL; j.right = s.dR; j.out = s.dO;
      j.left_frame_stride = fl; j.right_frame_stride = fr; j.out_frame_stride = fo;
      j.lstride = lstride; j.rstride = rstride; j.ostride = ostride;
      j.n = cnt; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp;
      j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.ox = 0; j.oy = 0;
      j.raw = (flags & SADHIP_FLAG_RAW) ? 1 : 0; j.row0 = 0;
      HIPTRY(hipStreamWaitEvent(c->s_kern, s.up, 0));
      HIPTRY(hipEventRecord(s.k0, c->s_kern));
      HIPTRY(dispatch(j, flags, c->s_kern));
      HIPTRY(hipEventRecord(s.k1, c->s_kern));
#ifdef SADHIP_TUNING
      // diagnostics build only (tests/test_parity_gpu.py): fail after the kernel of group SADHIP_TUNE_FAIL_GROUP has been
      // queued, with uploads, kernels and downloads of earlier groups still in flight -- the exit below must drain them
      if (const char* fg = getenv("SADHIP_TUNE_FAIL_GROUP"))
        if (atoi(fg) == g) return fail(SADHIP_ERR_HIP, "injected failure after the kernel of group %d (diagnostics build)", g);
#endif
      HIPTRY(hipStreamWaitEvent(c->s_down, s.k1, 0));
      if (int st = download(c->s_down, s.dO, i0, cnt)) return st;
      HIPTRY(hipEventRecord(s.down, c->s_down));
    }
    for (int g = ng > S ? ng - S : 0; g < ng; ++g) {            // the groups still in flight
      sadhip_ctx::GroupSlot& s = c->grp[g % S];
      HIPTRY(hipEventSynchronize(s.down));
      float ms = 0.f;
      HIPTRY(hipEventElapsedTime(&ms, s.k0, s.k1));
      *kms += ms;
    }
    return SADHIP_OK;
  };
  const int st = body();
  // nothing from / to the caller's memory stays in flight, whatever happened
  for (hipStream_t q : {c->s_up, c->s_kern, c->s_down}) (void)hipStreamSynchronize(q);
  return st;
}

extern "C" int sadhip_disparity_batch_u8(sadhip_ctx* ctx, int n,
                                         const uint8_t* const* lefts, const uint8_t* const* rights,
                                         int w, int h, int lstride, int rstride,
                                         int block, int maxdisp, unsigned flags,
                                         uint8_t* const* outs, int ostride)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  if (n < 0) return fail(SADHIP_ERR_BATCH, "n=%d", n);
  if (n == 0) return SADHIP_OK;
  if (!lefts || !rights || !outs) return fail(SADHIP_ERR_NULL, "lefts/rights/outs is NULL");
  for (int i = 0; i < n; ++i)
    if (!lefts[i] || !rights[i] || !outs[i]) return fail(SADHIP_ERR_NULL, "frame %d has a NULL pointer", i);
  if (int st = check_params(w, h, lstride, rstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (ostride < w) return fail(SADHIP_ERR_STRIDE, "ostride=%d w=%d", ostride, w);
  if (w > ctx->max_w || h > ctx->max_h) return fail(SADHIP_ERR_DIMS, "%dx%d exceeds context capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const auto t0 = clk::now();
  float kms = 0.f;
  {
    // every frame in page-locked memory this library knows, frames fit the device planes at the caller's strides
    bool pinned = (size_t)lstride * h <= ctx->plane && (size_t)rstride * h <= ctx->plane &&
                  (size_t)ostride * h <= ctx->plane;
    const size_t cl = (size_t)lstride * (h - 1) + w, cr = (size_t)rstride * (h - 1) + w, co = (size_t)ostride * (h - 1) + w;
    for (int i = 0; i < n && pinned; ++i)
      pinned = is_pinned(lefts[i], cl) && is_pinned(rights[i], cr) && is_pinned(outs[i], co);
    if (pinned) {
      if (int st = batch_pinned(ctx, n, lefts, rights, w, h, lstride, rstride, block, maxdisp, flags, outs, ostride, &kms)) return st;
      ctx->last_kernel_ms = kms;
      ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
      return SADHIP_OK;
    }
  }
  // software pipeline over the kSlots slots: while earlier frames are on the bus / the GPU, the host stages and
  // enqueues the next frame on a free slot, then drains the oldest (two slots left the host waiting for the GPU
  // 75 us per pair: profiles/r04_ab_hostcopy_2slots.txt against r04_ab_hostcopy.txt).
#ifdef SADHIP_TUNING
  static const bool seq = getenv("SADHIP_BATCH_SEQ") != nullptr;      // tuning aid: no cross-frame overlap
#else
  constexpr bool seq = false;
#endif
  if (seq) {
    for (int i = 0; i < n; ++i) {
      route_output(ctx, ctx->slot[0], outs[i], ostride, w, h);
      int st = enqueue_frame(ctx, ctx->slot[0], lefts[i], rights[i], w, h, lstride, rstride, block, maxdisp, flags, 0, 0, w, h, 0, h);
      if (st == SADHIP_OK) st = finish_frame(ctx, ctx->slot[0], w, 0, 0, w, h, &kms);
      if (st != SADHIP_OK) { (void)hipStreamSynchronize(ctx->slot[0].stream); return st; }      // nothing of the caller's stays in flight
    }
    ctx->last_kernel_ms = kms;
    ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
    return SADHIP_OK;
  }
  constexpr int S = sadhip_ctx::kSlots;
  // the frames of a batch run on the per-direction streams (Slot::up / kern / down), chained by events per slot
  if (int st = ensure_dir_streams(ctx)) return st;
  struct Route {
    sadhip_ctx* c;
    explicit Route(sadhip_ctx* cc) : c(cc)
    {
#ifdef SADHIP_TUNING
      if (getenv("SADHIP_TUNE_SLOTSTREAMS")) return;          // A/B: every slot on its own stream, as before
#endif
      for (auto& s : c->slot) { s.up = c->s_up; s.kern = c->s_kern; s.down = c->s_down; }
    }
    ~Route() { for (auto& s : c->slot) s.up = s.kern = s.down = s.stream; }
  } route(ctx);
  auto drain = [&] {                                       // error exit: nothing of the caller's stays in flight
    for (hipStream_t st : {ctx->s_up, ctx->s_kern, ctx->s_down}) (void)hipStreamSynchronize(st);
    for (auto& s : ctx->slot) (void)hipStreamSynchronize(s.stream);
  };
  for (int i = 0; i < n + S - 1; ++i) {
    if (i < n) {
      route_output(ctx, ctx->slot[i % S], outs[i], ostride, w, h);
      if (int st = enqueue_frame(ctx, ctx->slot[i % S], lefts[i], rights[i], w, h, lstride, rstride,
                                 block, maxdisp, flags, 0, 0, w, h, 0, h)) {
        drain();
        return st;
      }
    }
    const int d = i - (S - 1);
    if (d >= 0) {
      if (int st = finish_frame(ctx, ctx->slot[d % S], w, 0, 0, w, h, &kms)) {
        drain();
        return st;
      }
    }
  }
  ctx->last_kernel_ms = kms;
  ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return SADHIP_OK;
}

extern "C" int sadhip_disparity_dev_u8(sadhip_ctx* ctx, int n,
                                       const uint8_t* d_left, const uint8_t* d_right,
                                       size_t left_frame_stride, size_t right_frame_stride,
                                       int w, int h, int lstride, int rstride,
                                       int block, int maxdisp, unsigned flags,
                                       uint8_t* d_out, size_t out_frame_stride, int ostride,
                                       void* hip_stream)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  if (n < 0) return fail(SADHIP_ERR_BATCH, "n=%d", n);
  if (n == 0) return SADHIP_OK;
  if (!d_left || !d_right || !d_out) return fail(SADHIP_ERR_NULL, "d_left/d_right/d_out is NULL");
  if (int st = check_params(w, h, lstride, rstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (ostride < w) return fail(SADHIP_ERR_STRIDE, "ostride=%d w=%d", ostride, w);
  if (n > 1) {
    const size_t need_l = (size_t)(h - 1) * lstride + w, need_r = (size_t)(h - 1) * rstride + w,
                 need_o = (size_t)(h - 1) * ostride + w;
    if (left_frame_stride < need_l || right_frame_stride < need_r || out_frame_stride < need_o)
      return fail(SADHIP_ERR_STRIDE, "frame strides (%zu,%zu,%zu) smaller than a frame", left_frame_stride,
                  right_frame_stride, out_frame_stride);
  }
  // no ctx mutex: pure enqueue on the caller's stream, no context buffers touched
  HIPTRY(hipSetDevice(ctx->device));
  Job j{};
  j.left = d_left; j.right = d_right; j.out = d_out;
  j.left_frame_stride = left_frame_stride; j.right_frame_stride = right_frame_stride;
  j.out_frame_stride = out_frame_stride;
  j.lstride = lstride; j.rstride = rstride; j.ostride = ostride;
  j.n = n; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp;
  j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.ox = 0; j.oy = 0;
  j.raw = (flags & SADHIP_FLAG_RAW) ? 1 : 0; j.row0 = 0;
  hipStream_t s = (hipStream_t)hip_stream;   // NULL = HIP's null stream (torch's default stream)
  HIPTRY(dispatch(j, flags, s));
  return SADHIP_OK;
}

// Runs body(); on any failure waits for the stream, so that no asynchronous copy from / to the
// caller's buffers is still in flight when the entry point returns (header: host pointers are
// never retained after return -- the cgo rule).
template <class F>
static int synced_on_error(hipStream_t s, F&& body)
{
  const int st = body();
  if (st != SADHIP_OK && s) (void)hipStreamSynchronize(s);
  return st;
}

// ------------------------------------------------------------------------------------------
// SumAbsoluteDifferences queries
// ------------------------------------------------------------------------------------------
extern "C" int sadhip_sad_windows_u8(sadhip_ctx* ctx,
                                     const uint8_t* left, int lw, int lh, int lstride,
                                     const uint8_t* right, int rw, int rh, int rstride,
                                     int block, int n, const int32_t* queries, int32_t* out_sad)
{
  if (!ctx || !left || !right) return fail(SADHIP_ERR_NULL, "ctx/left/right is NULL");
  if (n < 0) return fail(SADHIP_ERR_BATCH, "n=%d", n);
  if (lw <= 0 || lh <= 0 || rw <= 0 || rh <= 0) return fail(SADHIP_ERR_DIMS, "left %dx%d right %dx%d", lw, lh, rw, rh);
  if (lstride < lw || rstride < rw) return fail(SADHIP_ERR_STRIDE, "lstride=%d rstride=%d", lstride, rstride);
  if (block < 1 || block > SADHIP_MAX_BLOCK) return fail(SADHIP_ERR_BLOCK, "block=%d", block);
  if (n == 0) return SADHIP_OK;
  if (!queries || !out_sad) return fail(SADHIP_ERR_NULL, "queries/out_sad is NULL");
  // The reference indexes Pix with unchecked offsets and panics when a window starts outside
  // the left image (sad.go:228,234); reject such queries instead.
  for (int i = 0; i < n; ++i) {
    const int lx = queries[4 * i], ly = queries[4 * i + 1];
    if (lx < 0 || ly < 0 || lx >= lw || ly >= lh)
      return fail(SADHIP_ERR_ARG, "query %d: left centre (%d,%d) outside the %dx%d left image", i, lx, ly, lw, lh);
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const size_t lbytes = (size_t)lw * lh, rbytes = (size_t)rw * rh;
  const size_t qbytes = (size_t)n * 16, obytes = (size_t)n * 4;
  const size_t off_r = align_up(lbytes, 256), off_q = off_r + align_up(rbytes, 256), off_o = off_q + align_up(qbytes, 256);
  const size_t total = off_o + align_up(obytes, 256);
  if (int st = ensure_scratch(ctx, total, total)) return st;
  uint8_t* hs = (uint8_t*)ctx->h_scratch; uint8_t* ds = (uint8_t*)ctx->d_scratch;
  for (int r = 0; r < lh; ++r) memcpy(hs + (size_t)r * lw, left + (size_t)r * lstride, (size_t)lw);
  for (int r = 0; r < rh; ++r) memcpy(hs + off_r + (size_t)r * rw, right + (size_t)r * rstride, (size_t)rw);
  memcpy(hs + off_q, queries, qbytes);
  hipStream_t s = ctx->slot[0].stream;
  return synced_on_error(s, [&]() -> int {
    HIPTRY(hipMemcpyAsync(ds, hs, off_o, hipMemcpyHostToDevice, s));
    HIPTRY(launch_sad_windows(ds, lw, lh, lw, ds + off_r, rw, rh, rw, block, n,
                              (const int32_t*)(ds + off_q), (int32_t*)(ds + off_o), s));
    HIPTRY(hipMemcpyAsync(hs + off_o, ds + off_o, obytes, hipMemcpyDeviceToHost, s));
    HIPTRY(hipStreamSynchronize(s));
    memcpy(out_sad, hs + off_o, obytes);
    return SADHIP_OK;
  });
}

// ------------------------------------------------------------------------------------------
// gray conversion and the fused RGBA entry point
// ------------------------------------------------------------------------------------------
static int check_pix(const uint8_t* pix, int channels, int w, int h, int stride)
{
  if (!pix) return fail(SADHIP_ERR_NULL, "pix is NULL");
  if (channels != 3 && channels != 4) return fail(SADHIP_ERR_ARG, "channels=%d (3 or 4)", channels);
  if (w <= 0 || h <= 0) return fail(SADHIP_ERR_DIMS, "w=%d h=%d", w, h);
  if ((long long)stride < (long long)w * channels) return fail(SADHIP_ERR_STRIDE, "stride=%d < w*channels", stride);
  return SADHIP_OK;
}

// one decoded picture (w x h pixels of `channels` bytes) -> device rows; returns the device row stride in *dstride.
// Page-locked memory this library knows: one linear copy, the device rows keep the caller's stride (dev holds cap
// bytes).  Anything else: through the page-locked plane `stage` (upload_staged), compact rows of prow bytes.
static int upload_pix(sadhip_ctx* c, hipStream_t st, uint8_t* dev, size_t cap, uint8_t* stage, size_t prow,
                      const uint8_t* pix, int channels, int w, int h, int stride, size_t* dstride)
{
  const size_t wbytes = (size_t)w * channels, span = (size_t)(h - 1) * stride + wbytes;
  if ((size_t)stride * h <= cap && is_pinned(pix, span)) {
    HIPTRY(hipMemcpyAsync(dev, pix, span, hipMemcpyHostToDevice, st));
    *dstride = (size_t)stride;
    return SADHIP_OK;
  }
  *dstride = prow;
  return upload_staged(c, st, dev, stage, prow, pix, (size_t)stride, wbytes, 0, h);
}

extern "C" int sadhip_gray_u8(sadhip_ctx* ctx, const uint8_t* pix, int channels,
                              int w, int h, int stride, uint8_t* gray, int gstride)
{
  if (!ctx || !gray) return fail(SADHIP_ERR_NULL, "ctx/gray is NULL");
  if (int st = check_pix(pix, channels, w, h, stride)) return st;
  if (gstride < w) return fail(SADHIP_ERR_STRIDE, "gstride=%d w=%d", gstride, w);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const size_t prow = align_up((size_t)w * channels, 16);
  const size_t pcap = align_up(std::max(prow, (size_t)stride) * h, 256);      // the picture at either row stride
  const size_t grow = align_up((size_t)w, 16), gbytes = align_up(grow * h, 256);
  if (int st = ensure_scratch(ctx, pcap + gbytes, pcap + gbytes)) return st;
  uint8_t* ds = (uint8_t*)ctx->d_scratch; uint8_t* hs = (uint8_t*)ctx->h_scratch;
  hipStream_t s = ctx->slot[0].stream;
  return synced_on_error(s, [&]() -> int {
    size_t dstride = 0;
    if (int st = upload_pix(ctx, s, ds, pcap, hs, prow, pix, channels, w, h, stride, &dstride)) return st;
    HIPTRY(launch_gray(ds, channels, w, h, (int)dstride, ds + pcap, (int)grow, s));
    if (is_pinned(gray, (size_t)(h - 1) * gstride + w)) {
      HIPTRY(hipMemcpy2DAsync(gray, (size_t)gstride, ds + pcap, grow, (size_t)w, h, hipMemcpyDeviceToHost, s));
      HIPTRY(hipStreamSynchronize(s));
      return SADHIP_OK;
    }
    HIPTRY(hipMemcpyAsync(hs + pcap, ds + pcap, grow * h, hipMemcpyDeviceToHost, s));
    HIPTRY(hipStreamSynchronize(s));
    copy_rows(ctx, RowCopy{gray, (size_t)gstride, hs + pcap, grow, (size_t)w, false}, h);
    return SADHIP_OK;
  });
}

// RGBA/RGB pair in, disparity out.  One upload per image into device scratch, gray conversion into the slot's device
// planes, disparity kernel, one download: no host gray plane and no second H2D.
extern "C" int sadhip_disparity_rgba_u8(sadhip_ctx* ctx,
                                        const uint8_t* left_pix, const uint8_t* right_pix, int channels,
                                        int w, int h, int lstride, int rstride,
                                        int block, int maxdisp, unsigned flags,
                                        uint8_t* out, int ostride)
{
  if (!ctx || !out) return fail(SADHIP_ERR_NULL, "ctx/out is NULL");
  if (int st = check_pix(left_pix, channels, w, h, lstride)) return st;
  if (int st = check_pix(right_pix, channels, w, h, rstride)) return st;
  if (int st = check_params(w, h, w, w, block, maxdisp)) return st;
  if (int st = check_flags(flags, SADHIP_FLAG_HOST_GRAY)) return st;
  if (ostride < w) return fail(SADHIP_ERR_STRIDE, "ostride=%d w=%d", ostride, w);
  if (w > ctx->max_w || h > ctx->max_h) return fail(SADHIP_ERR_DIMS, "%dx%d exceeds context capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPTRY(hipSetDevice(ctx->device));
  const auto t0 = clk::now();
  if (flags & SADHIP_FLAG_HOST_GRAY) {
    // The caller's choice (SURVEY 8 f-1 "offer both"): the copy threads, which read every byte of the pictures anyway, write
    // LUMINANCE rows into the slot's page-locked planes (host/gray_rows.h: the same integer formula as k_gray, bit for bit)
    // instead of copies of the pictures -- a quarter of the bytes staged and sent over the bus (1080p RGBA pair: 16.6 MB ->
    // 4.1 MB).  From there on the call is sadhip_disparity_u8's: the left plane's transfer overlaps the right picture's conversion.
    Slot& s = ctx->slot[0];
    float kms = 0.f;
    const unsigned kflags = flags & ~SADHIP_FLAG_HOST_GRAY;
    route_output(ctx, s, out, ostride, w, h);
    const int st = synced_on_error(s.stream, [&]() -> int {
      if (int e = ensure_stage(ctx, s)) return e;
      const size_t pitch = pitch_for(w);
      const uint8_t* pix[2] = {left_pix, right_pix};
      const int strides[2] = {lstride, rstride};
      uint8_t* host[2] = {s.hL, s.hR};
      uint8_t* dev[2] = {s.dL, s.dR};
      for (int k = 0; k < 2; ++k) {
        copy_rows(ctx, RowCopy{host[k], pitch, pix[k], (size_t)strides[k], (size_t)w * channels, false, channels, w}, h);
        HIPTRY(hipMemcpyAsync(dev[k], host[k], (size_t)(h - 1) * pitch + w, hipMemcpyHostToDevice, s.stream));
      }
      Job j{};
      j.left = s.dL; j.right = s.dR; j.out = s.dO;
      j.left_frame_stride = j.right_frame_stride = j.out_frame_stride = ctx->plane;
      j.lstride = j.rstride = j.ostride = (int)pitch;
      j.n = 1; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp;
      j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.ox = 0; j.oy = 0;
      j.raw = (kflags & SADHIP_FLAG_RAW) ? 1 : 0; j.row0 = 0;
      HIPTRY(hipEventRecord(s.k0, s.stream));
      HIPTRY(dispatch(j, kflags, s.stream));
      HIPTRY(hipEventRecord(s.k1, s.stream));
      if (int e = enqueue_download(ctx, s, w, 0, w, h)) return e;
      return finish_frame(ctx, s, w, 0, 0, w, h, &kms);
    });
    if (st != SADHIP_OK) return st;
    ctx->last_kernel_ms = kms;
    ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
    return SADHIP_OK;
  }
  const size_t prow = align_up((size_t)w * channels, 16);
  const size_t pcap = align_up(std::max({prow, (size_t)lstride, (size_t)rstride}) * h, 256);
  if (int st = ensure_scratch(ctx, 2 * pcap, 2 * pcap)) return st;
  uint8_t* ds = (uint8_t*)ctx->d_scratch; uint8_t* hs = (uint8_t*)ctx->h_scratch;
  Slot& s = ctx->slot[0];
  float kms = 0.f;
  route_output(ctx, s, out, ostride, w, h);
  const int st = synced_on_error(s.stream, [&]() -> int {
    size_t dls = 0, drs = 0;
    if (int e = upload_pix(ctx, s.stream, ds, pcap, hs, prow, left_pix, channels, w, h, lstride, &dls)) return e;
    if (int e = upload_pix(ctx, s.stream, ds + pcap, pcap, hs + pcap, prow, right_pix, channels, w, h, rstride, &drs)) return e;
    HIPTRY(hipEventRecord(s.k0, s.stream));
    const size_t gpitch = pitch_for(w);
    HIPTRY(launch_gray(ds, channels, w, h, (int)dls, s.dL, (int)gpitch, s.stream));
    HIPTRY(launch_gray(ds + pcap, channels, w, h, (int)drs, s.dR, (int)gpitch, s.stream));
    Job j{};
    j.left = s.dL; j.right = s.dR; j.out = s.dO;
    j.left_frame_stride = j.right_frame_stride = j.out_frame_stride = ctx->plane;
    j.lstride = j.rstride = j.ostride = (int)gpitch;
    j.n = 1; j.w = w; j.h = h; j.block = block; j.maxdisp = maxdisp;
    j.x0 = 0; j.y0 = 0; j.x1 = w; j.y1 = h; j.ox = 0; j.oy = 0;
    j.raw = (flags & SADHIP_FLAG_RAW) ? 1 : 0; j.row0 = 0;
    HIPTRY(dispatch(j, flags, s.stream));
    HIPTRY(hipEventRecord(s.k1, s.stream));
    if (int e = enqueue_download(ctx, s, w, 0, w, h)) return e;
    return finish_frame(ctx, s, w, 0, 0, w, h, &kms);
  });
  if (st != SADHIP_OK) return st;
  ctx->last_kernel_ms = kms;
  ctx->last_total_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return SADHIP_OK;
}

extern "C" int sadhip_last_timing(sadhip_ctx* ctx, float* kernel_ms, float* total_ms)
{
  if (!ctx) return fail(SADHIP_ERR_NULL, "ctx is NULL");
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (kernel_ms) *kernel_ms = ctx->last_kernel_ms;
  if (total_ms) *total_ms = ctx->last_total_ms;
  return SADHIP_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-device pool: one context and one host thread per entry of
// devices[]; frame i of a batch goes to entry i mod G (the split north_star names); parameters
// are fanned out by the host; results land in the caller's buffers.  No device-to-device
// exchange exists on this path, so no collective is involved.
// ------------------------------------------------------------------------------------------
namespace {

struct PoolTask {
  int n = 0;
  const uint8_t* const* lefts = nullptr; const uint8_t* const* rights = nullptr; uint8_t* const* outs = nullptr;
  int w = 0, h = 0, lstride = 0, rstride = 0, block = 0, maxdisp = 0, ostride = 0;
  unsigned flags = 0;
};

// Where a device hangs on the host: PCI bus id -> /sys/bus/pci/devices/<id>/{numa_node, local_cpulist}.  Best effort,
// no dependency beyond sysfs: anything that cannot be read leaves the worker unpinned.
struct Placement {
  int device = -1, numa_node = -1;
  bool pinned = false;
  char pci[32] = "";
  char cpus[256] = "";                 // the CPUs the worker thread was pinned to (local_cpulist cut down to the process's affinity mask)
  cpu_set_t set;
  bool have_set = false;
};

static bool parse_cpulist(const char* txt, cpu_set_t* out)
{
  CPU_ZERO(out);
  bool any = false;
  const char* p = txt;
  while (*p) {
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    if (!*p) break;
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p || a < 0) return false;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1 || b < a) return false; p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, out); any = true; }
  }
  return any;
}

static void format_cpulist(const cpu_set_t* set, char* buf, size_t cap)
{
  size_t at = 0;
  buf[0] = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, set)) continue;
    int e = c;
    while (e + 1 < CPU_SETSIZE && CPU_ISSET(e + 1, set)) ++e;
    const int n = e > c ? snprintf(buf + at, cap - at, "%s%d-%d", at ? "," : "", c, e) : snprintf(buf + at, cap - at, "%s%d", at ? "," : "", c);
    if (n < 0 || (size_t)n >= cap - at) { buf[at] = 0; return; }
    at += (size_t)n;
    c = e;
  }
}

static Placement find_placement(int device)
{
  Placement pl;
  pl.device = device;
  char id[32] = "";
  if (hipDeviceGetPCIBusId(id, (int)sizeof(id), device) != hipSuccess) { (void)hipGetLastError(); return pl; }
  for (char* c = id; *c; ++c) *c = (char)tolower((unsigned char)*c);      // sysfs names are lower case
  snprintf(pl.pci, sizeof(pl.pci), "%s", id);
  char path[128], txt[512];
  auto slurp = [&](const char* leaf) -> bool {
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/%s", id, leaf);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(txt, 1, sizeof(txt) - 1, f);
    fclose(f);
    txt[n] = 0;
    return n > 0;
  };
  if (slurp("numa_node")) pl.numa_node = atoi(txt);
  cpu_set_t local, mine;
  if (slurp("local_cpulist") && parse_cpulist(txt, &local) && sched_getaffinity(0, sizeof(mine), &mine) == 0) {
    CPU_AND(&pl.set, &local, &mine);                             // never leave the CPUs this process was given (cpuset / cgroup)
    pl.have_set = CPU_COUNT(&pl.set) > 0;
  }
  return pl;
}

struct PoolWorker {
  sadhip_ctx* ctx = nullptr;
  Placement place;
  int index = 0, stride = 1;           // this worker's frames: index, index + stride, ...
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  const PoolTask* task = nullptr;      // posted by the caller, cleared by the worker
  bool quit = false, done = false, started = false;
  // sadhip_pool_host_alloc: page-locked memory allocated AND first-touched by this (pinned) thread
  size_t alloc_bytes = 0; void* alloc_ptr = nullptr; bool alloc_pending = false, alloc_done = false; int alloc_status = SADHIP_OK;
  int status = SADHIP_OK;
  int frames = 0;
  float kernel_ms = 0.f;
  char err[512] = "";
};

void pool_worker_main(PoolWorker* me)
{
  std::vector<const uint8_t*> L, R;
  std::vector<uint8_t*> O;
  // the thread that issues this device's copies and launches runs on the CPUs next to the device: on a two-socket host a
  // thread on the other socket queues every copy across the socket link (its runtime-internal staging too)
  if (me->place.have_set && pthread_setaffinity_np(pthread_self(), sizeof(me->place.set), &me->place.set) == 0) {
    std::lock_guard<std::mutex> lk(me->mu);
    me->place.pinned = true;
    format_cpulist(&me->place.set, me->place.cpus, sizeof(me->place.cpus));
  }
  { std::lock_guard<std::mutex> lk(me->mu); me->started = true; }
  me->cv.notify_all();
  for (;;) {
    std::unique_lock<std::mutex> lk(me->mu);
    me->cv.wait(lk, [&] { return me->task != nullptr || me->quit || me->alloc_pending; });
    if (me->quit) return;
    if (me->alloc_pending) {                                    // allocate + first-touch on this thread's node
      const size_t bytes = me->alloc_bytes;
      lk.unlock();
      void* p = nullptr;
      // the runtime places page-locked memory by the calling thread's CURRENT device (a fresh thread's is device 0): make it
      // this worker's before the allocation -- the memset below cannot move pages that are already locked
      int ast = SADHIP_OK;
      if (hipSetDevice(me->ctx->device) != hipSuccess) { (void)hipGetLastError(); ast = fail(SADHIP_ERR_HIP, "hipSetDevice(%d) failed", me->ctx->device); }
      if (ast == SADHIP_OK) ast = sadhip_host_alloc(bytes, &p);
      if (ast == SADHIP_OK) memset(p, 0, bytes);
      lk.lock();
      me->alloc_ptr = p; me->alloc_status = ast;
      if (ast != SADHIP_OK) snprintf(me->err, sizeof(me->err), "device entry %d: %s", me->index, g_err);
      me->alloc_pending = false; me->alloc_done = true;
      lk.unlock();
      me->cv.notify_all();
      continue;
    }
    const PoolTask* t = me->task;
    lk.unlock();
    int st = SADHIP_OK;
    float kms = 0.f;
    int frames = 0;
    try {
      L.clear(); R.clear(); O.clear();
      for (int i = me->index; i < t->n; i += me->stride) { L.push_back(t->lefts[i]); R.push_back(t->rights[i]); O.push_back(t->outs[i]); }
      frames = (int)L.size();
      if (frames > 0) {
        st = sadhip_disparity_batch_u8(me->ctx, frames, L.data(), R.data(), t->w, t->h, t->lstride, t->rstride,
                                       t->block, t->maxdisp, t->flags, O.data(), t->ostride);
        if (st == SADHIP_OK) (void)sadhip_last_timing(me->ctx, &kms, nullptr);
      }
    } catch (const std::bad_alloc&) {
      st = fail(SADHIP_ERR_NOMEM, "pool worker %d: out of host memory", me->index);
    } catch (...) {
      st = fail(SADHIP_ERR_ARG, "pool worker %d: unexpected exception", me->index);
    }
    lk.lock();
    me->status = st;
    me->frames = frames;
    me->kernel_ms = kms;
    if (st != SADHIP_OK) snprintf(me->err, sizeof(me->err), "device entry %d: %s", me->index, g_err);   // g_err of THIS thread
    me->task = nullptr;
    me->done = true;
    lk.unlock();
    me->cv.notify_all();
  }
}

}  // namespace

struct sadhip_pool {
  std::vector<std::unique_ptr<PoolWorker>> w;
  std::mutex call_mu;                  // one batch at a time per pool
  float last_wall_ms = 0.f, last_kernel_ms = 0.f;
};

static void destroy_pool(sadhip_pool* p)
{
  if (!p) return;
  for (auto& w : p->w) {
    if (w->th.joinable()) {
      { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
      w->cv.notify_all();
      w->th.join();
    }
    if (w->ctx) destroy_ctx(w->ctx);
  }
  delete p;
}

extern "C" int sadhip_pool_create(const int* devices, int n_devices, int max_w, int max_h, sadhip_pool** out)
{
  if (!out) return fail(SADHIP_ERR_NULL, "out is NULL");
  *out = nullptr;
  if (!devices) return fail(SADHIP_ERR_NULL, "devices is NULL");
  if (n_devices < 1 || n_devices > 64) return fail(SADHIP_ERR_ARG, "n_devices=%d (valid 1..64)", n_devices);
  sadhip_pool* p = nullptr;
  try {
    p = new sadhip_pool();
    for (int g = 0; g < n_devices; ++g) {
      std::unique_ptr<PoolWorker> w(new PoolWorker());
      w->index = g; w->stride = n_devices;
      const int st = sadhip_ctx_create(devices[g], max_w, max_h, 1, &w->ctx);
      if (st != SADHIP_OK) { destroy_pool(p); return st; }     // detail already in g_err
      w->place = find_placement(devices[g]);
      p->w.push_back(std::move(w));
    }
    for (auto& w : p->w) w->th = std::thread(pool_worker_main, w.get());
    for (auto& w : p->w) {                                     // placement is settled when create returns (sadhip_pool_describe)
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->started; });
    }
  } catch (const std::bad_alloc&) {
    destroy_pool(p);
    return fail(SADHIP_ERR_NOMEM, "pool allocation failed");
  } catch (const std::exception& e) {                          // std::system_error from std::thread
    destroy_pool(p);
    return fail(SADHIP_ERR_ARG, "pool threads: %s", e.what());
  }
  *out = p;
  return SADHIP_OK;
}

extern "C" int sadhip_pool_destroy(sadhip_pool* pool)
{
  if (!pool) return fail(SADHIP_ERR_NULL, "pool is NULL");
  destroy_pool(pool);
  return SADHIP_OK;
}

extern "C" int sadhip_pool_size(const sadhip_pool* pool, int* n_devices)
{
  if (!pool || !n_devices) return fail(SADHIP_ERR_NULL, "pool or n_devices is NULL");
  *n_devices = (int)pool->w.size();
  return SADHIP_OK;
}

extern "C" int sadhip_pool_set_copy_threads(sadhip_pool* pool, int n)
{
  if (!pool) return fail(SADHIP_ERR_NULL, "pool is NULL");
  for (auto& w : pool->w)
    if (int st = sadhip_ctx_set_copy_threads(w->ctx, n)) return st;
  return SADHIP_OK;
}

extern "C" int sadhip_pool_set_copy_spin_us(sadhip_pool* pool, int us)
{
  if (!pool) return fail(SADHIP_ERR_NULL, "pool is NULL");
  for (auto& w : pool->w)
    if (int st = sadhip_ctx_set_copy_spin_us(w->ctx, us)) return st;
  return SADHIP_OK;
}

extern "C" int sadhip_pool_host_alloc(sadhip_pool* pool, int entry, size_t bytes, void** out)
{
  if (!pool || !out) return fail(SADHIP_ERR_NULL, "pool or out is NULL");
  *out = nullptr;
  if (entry < 0 || entry >= (int)pool->w.size()) return fail(SADHIP_ERR_ARG, "entry %d out of range [0,%d)", entry, (int)pool->w.size());
  if (bytes == 0) return fail(SADHIP_ERR_ARG, "bytes=0");
  std::lock_guard<std::mutex> call(pool->call_mu);            // not while a batch is running on the pool
  PoolWorker& w = *pool->w[entry];
  { std::lock_guard<std::mutex> lk(w.mu); w.alloc_bytes = bytes; w.alloc_ptr = nullptr; w.alloc_done = false; w.alloc_pending = true; }
  w.cv.notify_all();
  std::unique_lock<std::mutex> lk(w.mu);
  w.cv.wait(lk, [&] { return w.alloc_done; });
  if (w.alloc_status != SADHIP_OK) return fail(w.alloc_status, "%s", w.err);
  *out = w.alloc_ptr;
  return SADHIP_OK;
}

extern "C" int sadhip_device_describe(int device, char* buf, int cap)
{
  if (!buf) return fail(SADHIP_ERR_NULL, "buf is NULL");
  if (cap < 1) return fail(SADHIP_ERR_ARG, "cap=%d", cap);
  buf[0] = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(SADHIP_ERR_NO_DEVICE, "no HIP device"); }
  if (device < 0 || device >= n) return fail(SADHIP_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
  Placement pl = find_placement(device);
  char cpus[256] = "-";
  if (pl.have_set) format_cpulist(&pl.set, cpus, sizeof(cpus));
  snprintf(buf, (size_t)cap, "device=%d pci=%s numa_node=%d cpus=%s\n", device, pl.pci[0] ? pl.pci : "?", pl.numa_node, cpus);
  return SADHIP_OK;
}

extern "C" int sadhip_pool_describe(const sadhip_pool* pool, char* buf, int cap)
{
  if (!pool || !buf) return fail(SADHIP_ERR_NULL, "pool or buf is NULL");
  if (cap < 1) return fail(SADHIP_ERR_ARG, "cap=%d", cap);
  int at = 0;
  buf[0] = 0;
  for (const auto& w : pool->w) {
    std::lock_guard<std::mutex> lk(w->mu);
    if (at < cap)
      at += snprintf(buf + at, (size_t)(cap - at), "entry=%d device=%d pci=%s numa_node=%d pinned=%d cpus=%s\n", w->index, w->place.device,
                     w->place.pci[0] ? w->place.pci : "?", w->place.numa_node, (int)w->place.pinned, w->place.pinned ? w->place.cpus : "-");
  }
  return SADHIP_OK;
}

extern "C" int sadhip_pool_batch_u8(sadhip_pool* pool, int n,
                                    const uint8_t* const* lefts, const uint8_t* const* rights,
                                    int w, int h, int lstride, int rstride,
                                    int block, int maxdisp, unsigned flags,
                                    uint8_t* const* outs, int ostride)
{
  if (!pool) return fail(SADHIP_ERR_NULL, "pool is NULL");
  if (n < 0) return fail(SADHIP_ERR_BATCH, "n=%d", n);
  if (n == 0) return SADHIP_OK;
  if (!lefts || !rights || !outs) return fail(SADHIP_ERR_NULL, "lefts/rights/outs is NULL");
  for (int i = 0; i < n; ++i)
    if (!lefts[i] || !rights[i] || !outs[i]) return fail(SADHIP_ERR_NULL, "frame %d has a NULL pointer", i);
  if (int st = check_params(w, h, lstride, rstride, block, maxdisp)) return st;
  if (int st = check_flags(flags)) return st;
  if (ostride < w) return fail(SADHIP_ERR_STRIDE, "ostride=%d w=%d", ostride, w);
  std::lock_guard<std::mutex> call(pool->call_mu);
  const auto t0 = clk::now();
  PoolTask t;
  t.n = n; t.lefts = lefts; t.rights = rights; t.outs = outs;
  t.w = w; t.h = h; t.lstride = lstride; t.rstride = rstride; t.block = block; t.maxdisp = maxdisp; t.ostride = ostride; t.flags = flags;
  for (auto& wk : pool->w) {                                   // fan out: same parameters to every device
    { std::lock_guard<std::mutex> lk(wk->mu); wk->done = false; wk->task = &t; }
    wk->cv.notify_all();
  }
  int status = SADHIP_OK;
  float kms = 0.f;
  for (auto& wk : pool->w) {                                   // every worker finishes before t goes out of scope
    std::unique_lock<std::mutex> lk(wk->mu);
    wk->cv.wait(lk, [&] { return wk->done; });
    if (wk->status != SADHIP_OK && status == SADHIP_OK) status = fail(wk->status, "%s", wk->err);
    kms += wk->kernel_ms;
  }
  pool->last_kernel_ms = kms;
  pool->last_wall_ms = std::chrono::duration<float, std::milli>(clk::now() - t0).count();
  return status;
}

extern "C" int sadhip_pool_last_timing(sadhip_pool* pool, float* kernel_ms_sum, float* wall_ms)
{
  if (!pool) return fail(SADHIP_ERR_NULL, "pool is NULL");
  std::lock_guard<std::mutex> call(pool->call_mu);
  if (kernel_ms_sum) *kernel_ms_sum = pool->last_kernel_ms;
  if (wall_ms) *wall_ms = pool->last_wall_ms;
  return SADHIP_OK;
}
