// stress_host_mirror.cpp -- threaded test of the C++ mirror's HOST logic (despair.cpp) over the
// test double fake_sadhip.cpp; built with -fsanitize=thread by `make -C steroscopic-hardware_amd/host tsan`
// and run by tests/test_host_mirror.py on the CPU.  Checks (VERDICT r1 "harden the boundary"):
//   1. concurrent RunSad callers with growing frame sizes + concurrent pipelines: no context is
//      destroyed while a call is in flight, every result is right;
//   2. pipelines on two devices at once keep their own contexts;
//   3. a device error inside the pipeline worker reaches AssembleDisparityMap as a despair::Error and
//      the feeder thread is released (no hang, no silently short map);
//   4. the device pool wrapper returns every frame;
//   5. left and right images of different sizes (the WebUI uploads them independently, pkg/camera/output.go:134-162):
//      RunSad and the pipeline worker hand each image over with its own size -- the right image is allocated at
//      exactly rw x rh bytes, so the asan build (make asan-host) turns any read past it into a failure.
#include "../../steroscopic-hardware_amd/host/despair.hpp"
#include "../../include/sadhip.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

extern "C" int fake_violations(void);
extern "C" int fake_created(void);
extern "C" int fake_calls(void);

using namespace despair;

static std::shared_ptr<Gray> image(int w, int h, int seed)
{
  auto g = std::make_shared<Gray>(w, h);
  for (size_t i = 0; i < g->Pix.size(); ++i) g->Pix[i] = (uint8_t)(i * 31 + seed * 7);
  return g;
}

static bool check_map(const Gray& got, const Gray& L, const Gray& R, int block)
{
  if (got.Rect.Dx() != L.Rect.Dx() || got.Rect.Dy() != L.Rect.Dy()) return false;      // the map has the LEFT image's size (sad.go:135,168)
  for (int y = 0; y < L.Rect.Dy(); ++y)
    for (int x = 0; x < L.Rect.Dx(); ++x) {
      const uint8_t r = (x < R.Rect.Dx() && y < R.Rect.Dy()) ? R.Pix[(size_t)y * R.Stride + x] : 0;
      if (got.Pix[(size_t)y * got.Stride + x] != (uint8_t)((L.Pix[(size_t)y * L.Stride + x] ^ r) + block)) return false;
    }
  return true;
}

static bool run_pipeline(int device, int w, int h, int block, int chunk_rows, int rw = 0, int rh = 0)
{
  auto L = image(w, h, 1), R = image(rw > 0 ? rw : w, rh > 0 ? rh : h, 2);
  Pipeline p = SetupConcurrentSAD(2, device);                    // capacity 4: the feeder really blocks
  const int chunks = (h + chunk_rows - 1) / chunk_rows;
  std::thread feeder([&] { for (int y = 0; y < h; y += chunk_rows) p.in->send(InputChunk{L, R, Rectangle{0, y, w, std::min(y + chunk_rows, h)}}); });
  bool ok = true;
  try { const Gray got = AssembleDisparityMap(*p.out, L->Rect, chunks); ok = check_map(got, *L, *R, block); }
  catch (const Error&) { ok = false; }
  feeder.join();
  p.in->close();
  return ok;
}

int main()
{
  std::atomic<int> bad{0};
  const int block = 9;
  SetDefaultParams(Parameters{block, 64});
  // ---- 1 + 2: concurrent callers, growing sizes, two devices ------------------------------------
  {
    std::vector<std::thread> ts;
    for (int t = 0; t < 4; ++t)
      ts.emplace_back([&, t] {
        for (int i = 0; i < 12; ++i) {
          const int w = 64 + 700 * ((i + t) % 4), h = 48 + 400 * ((i * 3 + t) % 4);   // up to 2164 x 1248: beyond the 1920x1080 default
          auto L = image(w, h, t), R = image(w, h, i);
          const Gray got = RunSad(*L, *R, block, 64, t & 1);
          if (!check_map(got, *L, *R, block)) bad.fetch_add(1);
        }
      });
    for (int t = 0; t < 2; ++t)
      ts.emplace_back([&, t] { for (int i = 0; i < 4; ++i) if (!run_pipeline(t, 300 + 900 * (i & 1) * 2, 200 + 500 * (i & 1) * 2, block, 7)) bad.fetch_add(1); });
    for (auto& th : ts) th.join();
  }
  if (bad.load()) { fprintf(stderr, "FAIL: %d wrong results in the concurrent phase\n", bad.load()); return 1; }
  if (fake_violations()) { fprintf(stderr, "FAIL: %d context lifetime violations\n", fake_violations()); return 1; }
  // ---- 3: error propagation ----------------------------------------------------------------------
  {
    SetDefaultParams(Parameters{block, 77});                     // the double fails every call with maxdisp 77
    auto L = image(320, 400, 1), R = image(320, 400, 2);
    Pipeline p = SetupConcurrentSAD(2, 0);
    std::thread feeder([&] { for (int y = 0; y < 400; y += 2) p.in->send(InputChunk{L, R, Rectangle{0, y, 320, y + 2}}); });   // 200 chunks into capacity 4
    bool threw = false; int status = 0;
    try { (void)AssembleDisparityMap(*p.out, L->Rect, 200); }
    catch (const Error& e) { threw = true; status = e.status; }
    feeder.join();                                               // must return: the worker drains the input channel
    p.in->close();
    if (!threw || status != SADHIP_ERR_HIP) { fprintf(stderr, "FAIL: pipeline error not propagated (threw=%d status=%d)\n", (int)threw, status); return 1; }
    bool threw2 = false;
    try { (void)RunSad(*L, *R, block, 77); } catch (const Error& e) { threw2 = e.status == SADHIP_ERR_HIP; }
    if (!threw2) { fprintf(stderr, "FAIL: RunSad error not thrown\n"); return 1; }
    SetDefaultParams(Parameters{block, 64});
    // a consumer that gives up early (closes nothing, just stops reading) must not wedge the worker forever:
    Pipeline q = SetupConcurrentSAD(2, 0);
    q.in->send(InputChunk{L, R, Rectangle{0, 0, 320, 2}});
    q.in->close();
    Gray part = AssembleDisparityMap(*q.out, Rectangle{0, 0, 320, 2}, 1);
    if (part.Pix[5] != (uint8_t)((L->Pix[5] ^ R->Pix[5]) + block)) { fprintf(stderr, "FAIL: single chunk\n"); return 1; }
    bool short_thrown = false;
    try { (void)AssembleDisparityMap(*q.out, Rectangle{0, 0, 320, 2}, 1); } catch (const Error& e) { short_thrown = e.status == SADHIP_ERR_ARG; }
    if (!short_thrown) { fprintf(stderr, "FAIL: early-closed channel accepted silently\n"); return 1; }
  }
  // ---- 4: device pool ----------------------------------------------------------------------------
  {
    DevicePool pool({0, 1, 0}, 640, 480);
    std::vector<std::shared_ptr<const Gray>> ls, rs;
    for (int i = 0; i < 7; ++i) { ls.push_back(image(200, 100, i)); rs.push_back(image(200, 100, i + 9)); }
    const std::vector<Gray> maps = pool.RunSadBatch(ls, rs, block, 64);
    if (maps.size() != 7 || pool.size() != 3) { fprintf(stderr, "FAIL: pool size\n"); return 1; }
    for (int i = 0; i < 7; ++i) if (!check_map(maps[i], *ls[i], *rs[i], block)) { fprintf(stderr, "FAIL: pool frame %d\n", i); return 1; }
    HostFrames hl(7, 200, 100), hr(7, 200, 100), ho(7, 200, 100);                  // the page-locked overload
    for (int i = 0; i < 7; ++i) { std::copy(ls[i]->Pix.begin(), ls[i]->Pix.end(), hl.frame(i)); std::copy(rs[i]->Pix.begin(), rs[i]->Pix.end(), hr.frame(i)); }
    pool.RunSadBatch(hl, hr, ho, block, 64);
    for (int i = 0; i < 7; ++i)
      if (!std::equal(maps[i].Pix.begin(), maps[i].Pix.end(), ho.frame(i))) { fprintf(stderr, "FAIL: pool page-locked frame %d\n", i); return 1; }
  }
  // ---- 5: left / right of different sizes ---------------------------------------------------------
  {
    const int sizes[][4] = {{320, 200, 300, 180}, {320, 200, 320, 150}, {320, 200, 17, 200}, {320, 200, 400, 260}, {64, 48, 1, 1}, {320, 200, 330, 190}};
    for (const auto& z : sizes) {
      auto L = image(z[0], z[1], 3), R = image(z[2], z[3], 4);
      const Gray got = RunSad(*L, *R, block, 64);
      if (!check_map(got, *L, *R, block)) { fprintf(stderr, "FAIL: RunSad %dx%d vs %dx%d\n", z[0], z[1], z[2], z[3]); return 1; }
      if (!run_pipeline(0, z[0], z[1], block, 13, z[2], z[3])) { fprintf(stderr, "FAIL: pipeline %dx%d vs %dx%d\n", z[0], z[1], z[2], z[3]); return 1; }
    }
  }
  if (fake_violations()) { fprintf(stderr, "FAIL: %d context lifetime violations\n", fake_violations()); return 1; }
  printf("OK contexts=%d calls=%d violations=%d\n", fake_created(), fake_calls(), fake_violations());
  return 0;
}
