// run_stream.cpp -- counterpart of the reference's examples/run.go + examples/run.stream.go:
// RunSad once, then the long-lived SetupConcurrentSAD pipeline fed with OutputCamera-style row
// bands (pkg/camera/output.go:172-187) for a few frames, printing the elapsed time per frame.
//   usage: run_stream <left.raw> <right.raw> <w> <h> <block> <maxdisp> <out.raw> [frames]
// Raw u8 gray in/out so the harness has no PNG dependency; tests/test_host_mirror.py drives it.
#include "despair.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
  if (argc < 8) { fprintf(stderr, "usage: %s left.raw right.raw w h block maxdisp out.raw [frames]\n", argv[0]); return 2; }
  const int w = atoi(argv[3]), h = atoi(argv[4]), block = atoi(argv[5]), maxdisp = atoi(argv[6]);
  const int frames = argc > 8 ? atoi(argv[8]) : 3;
  auto L = load(argv[1], w, h), R = load(argv[2], w, h);
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
      // capacity; a separate feeder cannot deadlock.
      std::thread feeder([&] {
        for (int y = 0; y < h; y += chunkSize) p.in->send(InputChunk{L, R, Rectangle{0, y, w, std::min(y + chunkSize, h)}});
      });
      got = AssembleDisparityMap(*p.out, L->Rect, numChunks);
      feeder.join();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("Elapsed time: %.3f ms\n", ms);                           // run.stream.go:60-62
    }
    p.in->close();
    if (got.Pix != once.Pix) { fprintf(stderr, "channel pipeline and RunSad disagree\n"); return 1; }
    std::ofstream(argv[7], std::ios::binary).write(reinterpret_cast<const char*>(got.Pix.data()), (std::streamsize)got.Pix.size());
    printf("SAD(100,50 vs 95,50) = %d\n", w > 110 && h > 60 ? SumAbsoluteDifferences(*L, *R, 100, 50, 95, 50, block) : -1);
  } catch (const Error& e) {
    fprintf(stderr, "despair error %d: %s\n", e.status, e.what());
    return 3;
  }
  return 0;
}
