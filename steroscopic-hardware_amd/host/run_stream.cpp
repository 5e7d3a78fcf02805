// run_stream.cpp -- counterpart of the reference's examples/run.go + examples/run.stream.go:
// RunSad once, then the long-lived SetupConcurrentSAD pipeline fed with OutputCamera-style row
// bands (pkg/camera/output.go:172-187) for a few frames, printing the elapsed time per frame.
//   usage: run_stream <left.raw> <right.raw> <w> <h> <block> <maxdisp> <out.raw> [frames] [--devices d0,d1,...]
// With --devices the frames additionally run as one batch over a DevicePool (sadhip_pool_*: one
// context + host thread per listed device, frame i on entry i mod G; an ordinal may repeat).
// Raw u8 gray in/out so the harness has no PNG dependency; tests/test_host_mirror.py drives it.
// Exit codes: 0 ok, 1 mismatch between the paths, 2 usage / unreadable input, 3 device error.
#include "despair.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <algorithm>

using namespace despair;

static std::shared_ptr<Gray> load(const char* path, int w, int h)
{
  auto g = std::make_shared<Gray>(w, h);
  std::ifstream f(path, std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(g->Pix.data()), (std::streamsize)g->Pix.size())) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  return g;
}

int main(int argc, char** argv)
{
  std::vector<int> devices;
  std::vector<char*> pos;
  for (int i = 0; i < argc; ++i) {
    if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
      for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t));
    } else pos.push_back(argv[i]);
  }
  if (pos.size() < 8) { fprintf(stderr, "usage: %s left.raw right.raw w h block maxdisp out.raw [frames] [--devices d0,d1,...]\n", argv[0]); return 2; }
  const int w = atoi(pos[3]), h = atoi(pos[4]), block = atoi(pos[5]), maxdisp = atoi(pos[6]);
  const int frames = pos.size() > 8 ? atoi(pos[8]) : 3;
  auto L = load(pos[1], w, h), R = load(pos[2], w, h);
  try {
    Gray once = RunSad(*L, *R, block, maxdisp);                        // examples/run.go:30
    const int numWorkers = 32;                                         // run.stream.go:18
    printf("Using %d workers (one device context)\n", numWorkers);
    Pipeline p = SetupConcurrentSAD(numWorkers);
    Gray got;
    for (int i = 0; i < frames; ++i) {                                 // run.stream.go:33
      const int chunkSize = std::max(1, h / (numWorkers * 4));         // run.stream.go:43
      const int numChunks = (h + chunkSize - 1) / chunkSize;
      const auto t0 = std::chrono::steady_clock::now();
      // feeder thread: the reference feeds from the consuming goroutine and relies on channel
      // capacity; a separate feeder cannot deadlock.  If the worker fails it keeps
      // draining the input channel, so this thread always finishes and join() returns.
      std::thread feeder([&] {
        for (int y = 0; y < h; y += chunkSize) p.in->send(InputChunk{L, R, Rectangle{0, y, w, std::min(y + chunkSize, h)}});
      });
      std::exception_ptr failure;
      try { got = AssembleDisparityMap(*p.out, L->Rect, numChunks); }
      catch (...) { failure = std::current_exception(); }
      feeder.join();
      if (failure) { p.in->close(); std::rethrow_exception(failure); }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("Elapsed time: %.3f ms\n", ms);                           // run.stream.go:60-62
    }
    p.in->close();
    if (got.Pix != once.Pix) { fprintf(stderr, "channel pipeline and RunSad disagree\n"); return 1; }
    if (!devices.empty()) {                                            // BASELINE config 4's shape on this host's devices
      DevicePool pool(devices, w, h);
      std::vector<std::shared_ptr<const Gray>> ls((size_t)frames, L), rs((size_t)frames, R);
      const auto t0 = std::chrono::steady_clock::now();
      const std::vector<Gray> maps = pool.RunSadBatch(ls, rs, block, maxdisp);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      for (const Gray& m : maps)
        if (m.Pix != once.Pix) { fprintf(stderr, "device pool and RunSad disagree\n"); return 1; }
      printf("Pool of %d contexts: %d frames in %.3f ms\n", pool.size(), frames, ms);
      // the same batch from page-locked frames (sadhip_host_alloc): asynchronous copies at bus speed
      HostFrames hl(frames, w, h), hr(frames, w, h), ho(frames, w, h);
      for (int i = 0; i < frames; ++i) {
        memcpy(hl.frame(i), L->Pix.data(), L->Pix.size()); memcpy(hr.frame(i), R->Pix.data(), R->Pix.size());
        memset(ho.frame(i), 0xEE, (size_t)w * h);
      }
      const auto t1 = std::chrono::steady_clock::now();
      pool.RunSadBatch(hl, hr, ho, block, maxdisp);
      const double ms1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      for (int i = 0; i < frames; ++i)
        if (memcmp(ho.frame(i), once.Pix.data(), once.Pix.size()) != 0) { fprintf(stderr, "device pool (page-locked frames) and RunSad disagree\n"); return 1; }
      printf("Pool of %d contexts, page-locked frames: %d frames in %.3f ms\n", pool.size(), frames, ms1);
    }
    std::ofstream(pos[7], std::ios::binary).write(reinterpret_cast<const char*>(got.Pix.data()), (std::streamsize)got.Pix.size());
    printf("SAD(100,50 vs 95,50) = %d\n", w > 110 && h > 60 ? SumAbsoluteDifferences(*L, *R, 100, 50, 95, 50, block) : -1);
  } catch (const Error& e) {
    fprintf(stderr, "despair error %d: %s\n", e.status, e.what());
    return 3;
  }
  return 0;
}
