// fake_sadhip.cpp -- TEST DOUBLE of the C ABI in include/sadhip.h, CPU only.  Test infrastructure:
// it exists so that the HOST logic of the C++ mirror (context leases, channels, error propagation,
// the device pool wrapper) can run under ThreadSanitizer in the build container, where there is no
// GPU and sanitizers are CPU-only.  It is linked into tests/host/stress_host_mirror only -- never
// into the product -- and computes a made-up function (NOT a disparity): out = (L ^ R) + block.
//   * contexts are never freed, so a call on a destroyed context or a destroy during a call is
//     detected and counted (fake_violations) instead of crashing;
//   * maxdisp == 77 injects SADHIP_ERR_HIP;
//   * every compute call sleeps a little to widen race windows.
#include "../../include/sadhip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

struct sadhip_ctx { std::atomic<int> alive{1}, in_call{0}; int device = 0, w = 0, h = 0; };
struct sadhip_pool { int n = 0; sadhip_ctx* ctx[64] = {}; };

static std::atomic<int> g_violations{0}, g_created{0}, g_destroyed{0}, g_calls{0};
static thread_local char g_err[256] = "";

extern "C" {
SADHIP_API int fake_violations(void) { return g_violations.load(); }
SADHIP_API int fake_created(void) { return g_created.load(); }
SADHIP_API int fake_destroyed(void) { return g_destroyed.load(); }
SADHIP_API int fake_calls(void) { return g_calls.load(); }
}

namespace {
struct InCall {
  sadhip_ctx* c;
  explicit InCall(sadhip_ctx* c) : c(c) {
    c->in_call.fetch_add(1);
    if (!c->alive.load()) g_violations.fetch_add(1);            // call on a destroyed context
    g_calls.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(300));
    if (!c->alive.load()) g_violations.fetch_add(1);            // destroyed while we were inside
  }
  ~InCall() { c->in_call.fetch_sub(1); }
};
int inject(int maxdisp)
{
  if (maxdisp == 77) { snprintf(g_err, sizeof(g_err), "injected device failure"); return SADHIP_ERR_HIP; }
  if (maxdisp < 1) { snprintf(g_err, sizeof(g_err), "maxdisp=%d", maxdisp); return SADHIP_ERR_MAXDISP; }
  return SADHIP_OK;
}
}  // namespace

extern "C" {
int sadhip_abi_version(void) { return SADHIP_ABI_VERSION; }
int sadhip_device_count(int* n) { if (n) *n = 4; return SADHIP_OK; }
const char* sadhip_status_string(int s) { return s == SADHIP_OK ? "ok" : "fake error"; }
const char* sadhip_last_error(void) { return g_err; }
int sadhip_last_kernel_kind(void) { return 0; }

int sadhip_ctx_create(int device, int max_w, int max_h, int, sadhip_ctx** out)
{
  if (device < 0 || device >= 4) { snprintf(g_err, sizeof(g_err), "device %d", device); return SADHIP_ERR_NO_DEVICE; }
  sadhip_ctx* c = new sadhip_ctx();                             // deliberately leaked on destroy (see header comment)
  c->device = device; c->w = max_w; c->h = max_h;
  g_created.fetch_add(1);
  *out = c;
  return SADHIP_OK;
}
int sadhip_ctx_destroy(sadhip_ctx* c)
{
  if (!c) return SADHIP_ERR_NULL;
  if (c->in_call.load() != 0) g_violations.fetch_add(1);        // destroy while a call is in flight
  if (c->alive.exchange(0) == 0) g_violations.fetch_add(1);     // double destroy
  g_destroyed.fetch_add(1);
  return SADHIP_OK;
}
int sadhip_ctx_device(const sadhip_ctx* c, int* d) { *d = c->device; return SADHIP_OK; }
int sadhip_ctx_set_copy_threads(sadhip_ctx*, int n) { return n >= 1 && n <= 32 ? SADHIP_OK : SADHIP_ERR_ARG; }
int sadhip_ctx_get_copy_threads(sadhip_ctx*, int* n) { *n = SADHIP_DEFAULT_COPY_THREADS; return SADHIP_OK; }

int sadhip_disparity_region_u8(sadhip_ctx* c, const uint8_t* L, const uint8_t* R, int w, int h, int ls, int rs,
                               int block, int maxdisp, unsigned, int x0, int y0, int x1, int y1, uint8_t* out)
{
  InCall g(c);
  if (w > c->w || h > c->h) { g_violations.fetch_add(1); return SADHIP_ERR_DIMS; }   // the lease gave us too small a context
  if (int st = inject(maxdisp)) return st;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) out[(size_t)(y - y0) * (x1 - x0) + (x - x0)] = (uint8_t)((L[(size_t)y * ls + x] ^ R[(size_t)y * rs + x]) + block);
  return SADHIP_OK;
}
// Left and right with their own sizes: reads R ONLY inside rw x rh (a binding that passed the wrong size or a too
// small buffer is caught by AddressSanitizer in the asan build); a missing right pixel counts as 0.
int sadhip_disparity_lr_u8(sadhip_ctx* c, const uint8_t* L, int lw, int lh, int ls, const uint8_t* R, int rw, int rh, int rs,
                           int block, int maxdisp, unsigned, int x0, int y0, int x1, int y1, uint8_t* out, int os)
{
  InCall g(c);
  if (lw > c->w || lh > c->h) { g_violations.fetch_add(1); return SADHIP_ERR_DIMS; }
  if (rw <= 0 || rh <= 0) return SADHIP_ERR_DIMS;
  if (int st = inject(maxdisp)) return st;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const uint8_t r = (x < rw && y < rh) ? R[(size_t)y * rs + x] : 0;
      out[(size_t)(y - y0) * os + (x - x0)] = (uint8_t)((L[(size_t)y * ls + x] ^ r) + block);
    }
  return SADHIP_OK;
}
int sadhip_disparity_u8(sadhip_ctx* c, const uint8_t* L, const uint8_t* R, int w, int h, int ls, int rs,
                        int block, int maxdisp, unsigned, uint8_t* out, int os)
{
  InCall g(c);
  if (w > c->w || h > c->h) { g_violations.fetch_add(1); return SADHIP_ERR_DIMS; }
  if (int st = inject(maxdisp)) return st;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) out[(size_t)y * os + x] = (uint8_t)((L[(size_t)y * ls + x] ^ R[(size_t)y * rs + x]) + block);
  return SADHIP_OK;
}
int sadhip_disparity_rgba_u8(sadhip_ctx* c, const uint8_t* L, const uint8_t* R, int ch, int w, int h, int ls, int rs,
                             int block, int maxdisp, unsigned, uint8_t* out, int os)
{
  InCall g(c);
  if (int st = inject(maxdisp)) return st;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) out[(size_t)y * os + x] = (uint8_t)((L[(size_t)y * ls + x * ch] ^ R[(size_t)y * rs + x * ch]) + block);
  return SADHIP_OK;
}
int sadhip_sad_windows_u8(sadhip_ctx* c, const uint8_t*, int, int, int, const uint8_t*, int, int, int, int, int n, const int32_t* q, int32_t* o)
{
  InCall g(c);
  for (int i = 0; i < n; ++i) o[i] = q[4 * i] + q[4 * i + 1];
  return SADHIP_OK;
}
int sadhip_gray_u8(sadhip_ctx* c, const uint8_t* pix, int ch, int w, int h, int stride, uint8_t* gray, int gs)
{
  InCall g(c);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) gray[(size_t)y * gs + x] = pix[(size_t)y * stride + x * ch];
  return SADHIP_OK;
}
int sadhip_last_timing(sadhip_ctx*, float* k, float* t) { if (k) *k = 0.1f; if (t) *t = 0.2f; return SADHIP_OK; }

int sadhip_host_alloc(size_t bytes, void** out) { *out = ::operator new(bytes); return SADHIP_OK; }
int sadhip_host_free(void* p) { ::operator delete(p); return SADHIP_OK; }

int sadhip_pool_create(const int* devices, int n, int w, int h, sadhip_pool** out)
{
  sadhip_pool* p = new sadhip_pool();
  p->n = n;
  for (int i = 0; i < n; ++i)
    if (int st = sadhip_ctx_create(devices[i], w, h, 1, &p->ctx[i])) return st;
  *out = p;
  return SADHIP_OK;
}
int sadhip_pool_destroy(sadhip_pool* p) { for (int i = 0; i < p->n; ++i) sadhip_ctx_destroy(p->ctx[i]); return SADHIP_OK; }
int sadhip_pool_size(const sadhip_pool* p, int* n) { *n = p->n; return SADHIP_OK; }
int sadhip_pool_set_copy_threads(sadhip_pool*, int n) { return n >= 1 && n <= 32 ? SADHIP_OK : SADHIP_ERR_ARG; }
int sadhip_pool_batch_u8(sadhip_pool* p, int n, const uint8_t* const* ls, const uint8_t* const* rs, int w, int h, int lst, int rst,
                         int block, int maxdisp, unsigned flags, uint8_t* const* outs, int os)
{
  for (int i = 0; i < n; ++i)
    if (int st = sadhip_disparity_u8(p->ctx[i % p->n], ls[i], rs[i], w, h, lst, rst, block, maxdisp, flags, outs[i], os)) return st;
  return SADHIP_OK;
}
}
