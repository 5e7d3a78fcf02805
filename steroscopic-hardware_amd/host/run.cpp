// run.cpp -- counterpart of the reference's examples/run.go: load the stereo pair, RunSad with the
// default parameters of that example (16, 64), save the disparity map.
//   usage: run [left.png right.png [out.png [block maxdisp]]] [--host-gray] [--frame]
// --host-gray: despair::SetHostGray(true) -- the gray conversion of an RGB(A) pair in libsadhip's copy threads while they
// stage the pictures (SADHIP_FLAG_HOST_GRAY) instead of on the device; same map.
// The decoded pictures stay in memory between decode, disparity and SavePNG: an
// RGB(A) pair is uploaded once per image and converted to gray on the device
// (sadhip_disparity_rgba_u8) -- no host gray plane, no $HOME/*.png round trip.  The timing line
// mirrors the "depth map generated" elapsed log of pkg/camera/output.go:200-204.
#include "despair.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char** argv)
{
  bool frame_mode = false;
  if (argc > 1 && std::string(argv[argc - 1]) == "--frame") { frame_mode = true; --argc; }
  if (argc > 1 && std::string(argv[argc - 1]) == "--host-gray") { despair::SetHostGray(true); --argc; }
  const char* left = argc > 2 ? argv[1] : "./testdata/L_00001.png";     // run.go:21
  const char* right = argc > 2 ? argv[2] : "./testdata/R_00001.png";    // run.go:26
  const char* out = argc > 3 ? argv[3] : "output.png";                  // run.go:31
  const int block = argc > 5 ? atoi(argv[4]) : 16, maxdisp = argc > 5 ? atoi(argv[5]) : 64;   // run.go:30
  try {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const despair::Picture L = despair::LoadPicture(left), R = despair::LoadPicture(right);
    const auto t1 = clk::now();
    int uploads = 0;
    // --frame: what patches/0003 makes OutputCamera call per frame -- RunSadImages with the parameters READ from the
    // process-wide ones (written here first, as POST /update-params would), nothing written back
    despair::Gray got;
    if (frame_mode) {
      despair::SetDefaultParams(despair::Parameters{block, maxdisp});
      got = despair::RunSadImages(L, R, despair::DefaultParams(), -1, &uploads);
    } else {
      got = despair::RunSad(L, R, block, maxdisp, -1, &uploads);
    }
    const auto t2 = clk::now();
    float kernel_ms = 0.f, call_ms = 0.f;
    despair::LastTiming(&kernel_ms, &call_ms);
    despair::SavePNG(out, got);
    const auto t3 = clk::now();
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    printf("%s: %dx%d disparity map, block %d, max disparity %d\n", out, got.Rect.Dx(), got.Rect.Dy(), block, maxdisp);
    printf("depth map generated elapsed=%.3fms decode=%.3fms disparity=%.3fms (device %.3fms, uploads %d) encode=%.3fms\n",
           ms(t0, t3), ms(t0, t1), ms(t1, t2), kernel_ms, uploads, ms(t2, t3));
  } catch (const despair::Error& e) {
    fprintf(stderr, "despair error %d: %s\n", e.status, e.what());    // run.go panics
    return 1;
  }
  return 0;
}
